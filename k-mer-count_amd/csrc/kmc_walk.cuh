// kmc_walk.cuh -- KMC_ALGO_WALK and the LR kernel (placeholder until the kernels land).
#pragma once
#include "kmc_device.cuh"
#include "../../include/kmc.h"

#define KMC_WALK_MAX_K 31
#define KMC_WALK_MAX_READ 512

static inline bool kmc_walk_supported(int, int, u64) { return false; }
static inline size_t kmc_walk_workspace_bytes(int, int) { return 256; }
static inline int kmc_walk_launch(hipStream_t, int, int, int, bool, const uint8_t*, const u64*, u64, u64, u64, void*, GTable) { return KMC_ERR_ARG; }
static inline int kmc_lr_launch(hipStream_t, int, const uint8_t*, const u64*, u64, u64, GTable) { return KMC_ERR_ARG; }
