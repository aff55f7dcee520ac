// kernels_attn_bf16_lazy.hip — the LAZY kernels of kernels_attn_bf16.hip as a translation unit of their own, compiled with
// -fno-slp-vectorize (Makefile; the reason is in that file's header).
#define HOLO_ATTN_LAZY_TU 1
#include "kernels_attn_bf16.hip"
