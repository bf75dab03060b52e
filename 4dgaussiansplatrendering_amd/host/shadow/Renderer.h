// Renderer.h — replaces the reference's 4DSplatRendering/Renderer.h: the class(es) it declares are provided by gs4d_compat.h over libgs4d.so.
// Copy this file over the reference's; everything that includes "Renderer.h" keeps compiling unchanged.
#pragma once
#include "gs4d_compat.h"
