// Software-pipelined attention forward (attention_pl.hip), dispatched from slh_attn_fwd (attention.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/sliders_hip.h"

// form = depth of the LDS ring (2 or 4); 128-query workgroups, D = 64, whole key tiles
bool slh_attn_pl_ok(const slh_attn_desc* d, int form);
int slh_attn_pl_blocks(const slh_attn_desc* d, int form);
int slh_attn_pl_launch(const slh_attn_desc* d, int form, hipStream_t s);
