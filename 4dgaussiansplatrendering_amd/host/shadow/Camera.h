// Camera.h — replaces the reference's 4DSplatRendering/Camera.h: the class(es) it declares are provided by gs4d_compat.h over libgs4d.so.
// Copy this file over the reference's; everything that includes "Camera.h" keeps compiling unchanged.
#pragma once
// the headers the reference's file of this name pulls in (the rest of the tree relies on them transitively)
#include <math.h>
#include "glm/glm.hpp"
#include "glm/gtc/matrix_transform.hpp"
#include "glm/gtx/rotate_vector.hpp"
#include "glm/gtx/vector_angle.hpp"
#include "gs4d_compat.h"
