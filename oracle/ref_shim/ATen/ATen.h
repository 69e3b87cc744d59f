#pragma once
#include "../cuda_shim.h"
namespace at { struct Half {}; struct Tensor; }
