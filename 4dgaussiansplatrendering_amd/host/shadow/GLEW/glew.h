// GLEW/glew.h — replaces Dependencies/GLEW/include/GLEW/glew.h: the GL types, enums and the handful of entry points the scene code
// calls itself (glGenBuffers, glBindBuffer, glBufferStorage, glBufferSubData, glBindBufferBase, glDeleteBuffers, Scenes.h:241-247,
// 321-325, 336, 220-224) are inline functions over libgs4d.so in gs4d_compat.h.  No OpenGL library is involved.
#pragma once
#include "gs4d_compat.h"
