// HIP counterpart of scanner/util/cuda.h as the reference GPU kernels use it
// (CU_CHECK(cudaSetDevice(id)), CUDA_PROTECT({...}); histogram_kernel_gpu.cpp:67-70).
#pragma once
#include <hip/hip_runtime_api.h>

#include "scanner/util/common.h"

#define HIP_CHECK(expr)                                                                       \
  do {                                                                                        \
    hipError_t e__ = (expr);                                                                  \
    LOG_IF(FATAL, e__ != hipSuccess) << "HIP error: " << hipGetErrorString(e__) << " (" #expr ")"; \
  } while (0)
#define HIP_PROTECT(block__) block__
