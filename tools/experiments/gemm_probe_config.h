// Probe build only: which tile of every persistent block (0 = its first, 2 = steady state) gemm_big_kernel<..., TRACE> stamps.
#pragma once
#ifndef AV_TRACE_TILE
#define AV_TRACE_TILE 2
#endif
