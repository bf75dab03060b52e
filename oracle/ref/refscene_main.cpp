// refscene_main.cpp — headless driver for the REFERENCE'S OWN scene classes (4DSplatRendering/Scenes.h, Splat.h, Scene.h, Utils.cpp,
// VDataParser.h — compiled unmodified, where they lie) running on libgs4d.so through the shadow headers (host/shadow/).  It does what
// Application.cpp:97-190 does around a scene — camera, clear colour, blend state, Update / Render / GUI — minus the window.
// Built only where the reference tree exists (oracle/Makefile target `refscene`, output oracle/_ref/refscene; the reference's sources
// are never copied).  tests/test_gpu_refscene.py runs it on the GPU box and checks the frame against the CPU checker.
//
//   refscene <scene> <out.rgba32f> <width> <height> [Label=value ...]      cwd must hold ../Objects/teapot.vdata (Scenes.h:231)
//   scene: linear | nonlinear | rotation | combined | broken | square | g2d | g3d      Label=value: scripted GUI widgets (imgui.h)
#include <GLEW/glew.h>
#include <GLFW/glfw3.h>
#include <stdlib.h>
#include <iostream>
#include <fstream>
#include <string>
#include <sstream>
#include <algorithm>
#include <chrono>
#include <functional>
#include <memory>
#include "Camera.h"
#include "Renderer.h"
#include "VertexBuffer.h"
#include "IndexBuffer.h"
#include "VertexArray.h"
#include "VertexBufferLayout.h"
#include "Shader.h"
#include "Geometry.h"
#include "glm/glm.hpp"
#include "glm/gtc/matrix_transform.hpp"
#include <glm/gtc/quaternion.hpp>
#include <glm/common.hpp>
#include <glm/gtx/matrix_decompose.hpp>
#include <glm/gtx/matrix_operation.hpp>
#include "Splat.h"
#include "imgui.h"
#include "Utils.h"
#include "radix_sort.hpp"
#include "BSPTree.h"
#include "ShareStorageBuffer.h"
#include "VDataParser.h"
#include "Scene.h"
#include "Scenes.h"

template <class S> static int run(const char* out, int W, int H, gs4d_ctx* ctx) {
    // Application.cpp:121-126: camera, clear colour, far plane
    Camera cam(W, H);
    Renderer renderer;
    glClearColor(0.18431373f, 0.20784314f, 0.25882353f, 1.0f);
    cam.SetFar(5000.0f);
    GLFWwindow window;
    window.keys[GLFW_KEY_M] = 1;                       // opens the scene menu, so that GUI() reaches its widgets
    {
        std::unique_ptr<Scene> scene = std::make_unique<S>(renderer, cam);
        scene->init();
        // one iteration of the frame loop, Application.cpp:145-182 (GUI() runs after Render() there; a second iteration would render
        // with what GUI() changed — here GUI() runs first so that ONE frame shows the scripted state)
        scene->Update(&window);
        scene->GUI();
        renderer.Clear();
        cam.HandleInput(&window);
        glBlendFunc(GL_SRC_ALPHA, GL_ONE_MINUS_SRC_ALPHA);
        glEnable(GL_BLEND); glDisable(GL_DEPTH_TEST);
        scene->Update(&window);
        scene->Render();
        std::vector<float> img((size_t)W * H * 4);
        gs4d::compat::Check(gs4d_read_pixels(ctx, img.data(), img.size() * 4), "gs4d_read_pixels");
        std::ofstream f(out, std::ios::binary);
        f.write(reinterpret_cast<const char*>(img.data()), (std::streamsize)(img.size() * 4));
        if (!f) { std::cerr << "refscene: cannot write " << out << "\n"; return 1; }
        scene->unload();
    }   // the scene's destructor deletes its GL names a second time (Scenes.h:220-224): tolerated
    const glm::vec3 p = cam.GetPosition();
    std::cout << "refscene: camera " << p.x << " " << p.y << " " << p.z << " orientation " << cam.orientation.x << " " << cam.orientation.y << " " << cam.orientation.z << "\n";
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 5) { std::cerr << "usage: refscene <scene> <out.rgba32f> <width> <height> [Label=value ...]\n"; return 2; }
    const std::string name = argv[1];
    const int W = atoi(argv[3]), H = atoi(argv[4]);
    for (int i = 5; i < argc; ++i) { const std::string a = argv[i]; const size_t eq = a.find('='); if (eq != std::string::npos) ImGui::Script()[a.substr(0, eq)] = atof(a.c_str() + eq + 1); }
    gs4d_ctx* ctx = nullptr;
    if (gs4d_create(0, W, H, &ctx) != GS4D_OK) { std::cerr << "gs4d_create: " << gs4d_last_error(nullptr) << "\n"; return 1; }
    gs4d::compat::MakeCurrent(ctx);
    int rc = 2;
    try {
        if (name == "linear") rc = run<Scenes::LinearMotion>(argv[2], W, H, ctx);
        else if (name == "nonlinear") rc = run<Scenes::NonLinearMotion>(argv[2], W, H, ctx);
        else if (name == "rotation") rc = run<Scenes::RotationMotion>(argv[2], W, H, ctx);
        else if (name == "combined") rc = run<Scenes::CombinedMotion>(argv[2], W, H, ctx);
        else if (name == "broken") rc = run<Scenes::BrokenMotion>(argv[2], W, H, ctx);
        else if (name == "square") rc = run<Scenes::SquareMotion>(argv[2], W, H, ctx);
        else if (name == "g2d") rc = run<Scenes::Gaussians2D>(argv[2], W, H, ctx);
        else if (name == "g3d") rc = run<Scenes::Gaussians3D>(argv[2], W, H, ctx);
        else std::cerr << "refscene: unknown scene " << name << "\n";
    } catch (const std::exception& e) { std::cerr << "refscene failed: " << e.what() << "\n"; rc = 1; }
    gs4d::compat::MakeCurrent(nullptr);
    gs4d_destroy(ctx);
    return rc;
}
