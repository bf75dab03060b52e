// radix_sort.hpp — replaces the reference's Dependencies/GPU_RADIX_SORT/radix_sort.hpp: the class(es) it declares are provided by gs4d_compat.h over libgs4d.so.
// Copy this file over the reference's; everything that includes "radix_sort.hpp" keeps compiling unchanged.
#pragma once
#include "gs4d_compat.h"
