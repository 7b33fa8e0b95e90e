// Attention backward (head_dim 64) - placeholder until the dQ / dK,dV kernels land.
#include "common.h"
#include "../../include/sliders_hip.h"

extern "C" int slh_attn_bwd(const slh_attn_bwd_desc* d, slh_stream_t stream) {
    (void)d; (void)stream;
    slh_set_error("slh_attn_bwd: not implemented yet");
    return -4;
}
