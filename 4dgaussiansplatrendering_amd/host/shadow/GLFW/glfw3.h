// GLFW/glfw3.h — replaces Dependencies/GLFW/include/GLFW/glfw3.h for the four input calls the path's host code makes
// (glfwGetKey, glfwSetInputMode, glfwSetCursorPos, glfwGetCursorPos: Camera.cpp:116-207, Scenes.h Update()).  A GLFWwindow here is
// just its input state (key table, cursor position) — set it from whatever window system, test or script drives the application.
#pragma once
#include "gs4d_compat.h"
