// link stubs for the device API so that the host layer can be exercised under ASan/UBSan on the CPU
#include "rtx.h"
extern "C" {
int rtx_create(int, rtx_ctx**) { return RTX_ERR_NO_DEVICE; }
void rtx_destroy(rtx_ctx*) {}
const char* rtx_last_error(rtx_ctx*) { return "stub"; }
int rtx_set_option(rtx_ctx*, int, int64_t) { return RTX_ERR_NO_DEVICE; }
int rtx_set_stream(rtx_ctx*, void*) { return RTX_ERR_NO_DEVICE; }
int rtx_set_materials(rtx_ctx*, const void*, uint32_t) { return RTX_ERR_NO_DEVICE; }
int rtx_add_mesh(rtx_ctx*, const void*, uint32_t, const uint32_t*, uint32_t, const uint32_t*, uint32_t*) { return RTX_ERR_NO_DEVICE; }
int rtx_add_instance(rtx_ctx*, uint32_t, const float*, uint32_t*) { return RTX_ERR_NO_DEVICE; }
int rtx_set_instance_transform(rtx_ctx*, uint32_t, const float*) { return RTX_ERR_NO_DEVICE; }
int rtx_commit_scene(rtx_ctx*) { return RTX_ERR_NO_DEVICE; }
int rtx_set_camera(rtx_ctx*, const float*, const float*) { return RTX_ERR_NO_DEVICE; }
int rtx_clear_accum(rtx_ctx*, uint32_t, uint32_t) { return RTX_ERR_NO_DEVICE; }
int rtx_render(rtx_ctx*, const rtx_params*) { return RTX_ERR_NO_DEVICE; }
int rtx_render_restir(rtx_ctx*, const rtx_params*) { return RTX_ERR_NO_DEVICE; }
int rtx_read_accum(rtx_ctx*, float*, size_t) { return RTX_ERR_NO_DEVICE; }
int rtx_read_srgb8(rtx_ctx*, uint8_t*, size_t) { return RTX_ERR_NO_DEVICE; }
int rtx_get_stats(rtx_ctx*, rtx_stats*) { return RTX_ERR_NO_DEVICE; }
int rtx_load_scene_cache(rtx_ctx*, const char*) { return RTX_ERR_NO_DEVICE; }
int rtx_read_layer(rtx_ctx*, uint32_t, uint32_t, uint32_t, uint8_t*, size_t) { return RTX_ERR_NO_DEVICE; }
}
