// exact_reference.hpp -- mode A device kernels (filled in below).
#pragma once
#include "device_common.hpp"
