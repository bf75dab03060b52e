// VertexBufferLayout.h — replaces the reference's 4DSplatRendering/VertexBufferLayout.h: the class(es) it declares are provided by gs4d_compat.h over libgs4d.so.
// Copy this file over the reference's; everything that includes "VertexBufferLayout.h" keeps compiling unchanged.
#pragma once
// the headers the reference's file of this name pulls in (the rest of the tree relies on them transitively)
#include <vector>
#include "gs4d_compat.h"
