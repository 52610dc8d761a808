// C ABI of libgpmpc_hip.so (include/gpmpc.h): host-side orchestration of the HIP kernels.
// One translation unit: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC gpmpc_api.hip
// The host code is split by concern into the api_*.inl files included below, in dependency order.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/gpmpc.h"
#include "chol_chain.hpp"
#include "chol_worker.hpp"
#include "em_kernels.hpp"
#include "gemm_f64_dma.hpp"
#include "gp_kernels.hpp"
#include "leaf64.hpp"
#include "train_native.hpp"
#include "vargemm_persist.hpp"

using namespace gpmpc;


#include "api_core.inl"
#include "api_factor.inl"
#include "api_handle.inl"
#include "api_fit.inl"
#include "api_predict.inl"
#include "api_rollout.inl"
#include "api_train.inl"
#include "api_lowlevel.inl"
