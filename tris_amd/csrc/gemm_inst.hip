// One operand-kind pair of the GEMM / implicit-GEMM family per translation unit: build.sh compiles this file seven times with
// -DTRIS_GEMM_AK=<A kind> -DTRIS_GEMM_BK=<B kind> (gemm_params.h enums), side by side.
#include "gemm_core.h"

#ifndef TRIS_GEMM_AK
#error "compile with -DTRIS_GEMM_AK=.. -DTRIS_GEMM_BK=.."
#endif
#define TRIS_CAT3(a, b, c) a##b##c
#define TRIS_RUN_NAME(a, b) TRIS_CAT3(tris_internal_run_cfg_, a, b)

// params: a GemmParams, cfg: a Cfg (same layouts in every unit: gemm_params.h)
extern "C" __attribute__((visibility("hidden"))) int TRIS_RUN_NAME(TRIS_GEMM_AK, TRIS_GEMM_BK)(const void* params, int batch, float* ws,
                                                                                              void* stream, const void* cfg, int mode) {
  return run_cfg<TRIS_GEMM_AK, TRIS_GEMM_BK>(*reinterpret_cast<const GemmParams*>(params), batch, ws, (hipStream_t)stream,
                                            *reinterpret_cast<const Cfg*>(cfg), mode);
}
