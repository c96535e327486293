// What ptk_group.hip needs from the single-GPU implementation in ptk_kernels.hip beyond the public C ABI.
#pragma once

#include <cstddef>
#include <string>

#include "../../include/ptk.h"

namespace ptk {

int ctx_device(const ptk_ctx* ctx);
float* ctx_accum(ptk_ctx* ctx);                  // the accumulation buffer in use (device memory, 4 floats per pixel)
size_t ctx_accum_pixels(const ptk_ctx* ctx);     // 0 before a camera is set
void ctx_set_error(ptk_ctx* ctx, const std::string& msg);
// ptk_upload_scene_ex that keeps the host-side copy of the scene whole, so that ctx_upload_like can hand the
// same flattened scene (and host-built tree) to the other GPUs' contexts; ctx_release_host_copy drops it.
int ctx_upload_keep_host(ptk_ctx* ctx, const ptk_scene_desc* scene, int32_t builder);
int ctx_upload_like(ptk_ctx* to, ptk_ctx* from);
void ctx_release_host_copy(ptk_ctx* ctx);
void ctx_drop_scene(ptk_ctx* ctx);                // frees the context's scene: later renders fail cleanly with "upload a scene first"

}  // namespace ptk
