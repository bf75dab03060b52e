// imgui.h — headless stand-in for Dependencies/IMGUI/imgui.h.  The scenes' GUI() bodies (out of scope of this path) are the only way
// the reference exposes its scene parameters (sort on/off, time, camera presets: Scenes.h:358-421), so the stand-in is SCRIPTABLE: a
// widget whose label is in ImGui::Script() takes its value from there (once), every other widget reports "not changed".  A headless
// driver sets e.g. Script()["Sort"] = 1, Script()["Time"] = 12.5, Script()["Cam_2"] = 1 and calls scene->GUI() as Application.cpp:166-182
// does.  An application that keeps its Dear ImGui window simply leaves the real imgui.h in place.
#pragma once
#include <string>
#include <unordered_map>
struct ImVec2 { float x, y; ImVec2(float a = 0, float b = 0) : x(a), y(b) {} };
struct ImVec4 { float x, y, z, w; ImVec4(float a = 0, float b = 0, float c = 0, float d = 0) : x(a), y(b), z(c), w(d) {} };
namespace ImGui {
inline std::unordered_map<std::string, double>& Script() { static thread_local std::unordered_map<std::string, double> s; return s; }
inline bool TakeScripted(const char* label, double& v) { auto& s = Script(); auto it = s.find(label); if (it == s.end()) return false; v = it->second; s.erase(it); return true; }
template <class... A> inline bool Begin(const char*, A&&...) { return true; }
inline void End() {}
inline void NewLine() {}
inline void SameLine(float = 0.0f, float = -1.0f) {}
template <class... A> inline void Text(const char*, A&&...) {}
template <class... A> inline bool Button(const char* label, A&&...) { double v; return TakeScripted(label, v) && v != 0.0; }
template <class... A> inline bool Checkbox(const char* label, bool* p, A&&...) { double v; if (!TakeScripted(label, v)) return false; *p = v != 0.0; return true; }
template <class... A> inline bool InputFloat(const char* label, float* p, A&&...) { double v; if (!TakeScripted(label, v)) return false; *p = (float)v; return true; }
template <class... A> inline bool SliderFloat(const char* label, float* p, A&&...) { double v; if (!TakeScripted(label, v)) return false; *p = (float)v; return true; }
template <class... A> inline bool InputInt(const char* label, int* p, A&&...) { double v; if (!TakeScripted(label, v)) return false; *p = (int)v; return true; }
template <class... A> inline bool DragFloat3(const char*, A&&...) { return false; }
template <class... A> inline bool DragFloat4(const char*, A&&...) { return false; }
template <class... A> inline bool ColorPicker4(const char*, A&&...) { return false; }
}
