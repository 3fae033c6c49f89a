// bf16 mixed-precision plan (net_bf16.hip) as seen by the C-ABI dispatch in net.hip.
#pragma once
#include "ursn_common.h"

struct ursn_bnet;
int bnet_query(const ursn_config* cfg, ursn_sizes* out);
int bnet_layer(const ursn_config* cfg, int64_t index, ursn_layer_info* out, int* n_layers);
int bnet_create(const ursn_config* cfg, float* params, float* grads, void* workspace, size_t workspace_bytes, ursn_bnet** out);
void bnet_destroy(ursn_bnet* n);
const ursn_sizes* bnet_sizes(const ursn_bnet* n);
float* bnet_metrics(ursn_bnet* n);
int bnet_param(const ursn_bnet* n, int64_t index, ursn_param_info* out);
// mode 0: forward + loss + backward (gradients accumulate); 1: forward + loss; 2: forward + softmax / ana labels
int bnet_step(ursn_bnet* n, const float* data, const float* label, const float* weight, int N, int mode, float* softmax_out,
              float* labels_out, hipStream_t s);
int bnet_tensor(const ursn_bnet* n, const char* name, void** ptr, int64_t* voxels, int32_t* channels, int32_t* cstride);
// per-launch HIP-event records (as ursn_profile_enable / ursn_profile_read of the fp32 plan) and the weight-gradient stream switch
int bnet_profile_enable(ursn_bnet* n, int on);
int bnet_profile_read(ursn_bnet* n, ursn_prof_rec* out, int64_t max_recs, int64_t* n_out);
int bnet_set_wgrad_overlap(ursn_bnet* n, int on);
