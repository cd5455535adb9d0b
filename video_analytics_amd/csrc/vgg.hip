// placeholder while the conv/FC kernels are being written (replaced in the next commit)
#include "va_internal.h"
extern "C" int va_vgg16_create(va_ctx*, int, int, int, int, const void* const*, const void* const*, const void* const*, const void* const*, const float*, const float*, void*, va_vgg16**) { va_set_error("va_vgg16: not built yet"); return VA_ERR_INVALID; }
extern "C" void va_vgg16_destroy(va_vgg16*) {}
extern "C" size_t va_vgg16_workspace_bytes(const va_vgg16*, int) { return 0; }
extern "C" int va_vgg16_forward(va_vgg16*, const void*, int, int, void*, void*, void*, void*, size_t, void*) { va_set_error("va_vgg16: not built yet"); return VA_ERR_INVALID; }
extern "C" int va_copy_first_layer(va_ctx*, const void*, int, int, void*, void*) { va_set_error("not built yet"); return VA_ERR_INVALID; }
extern "C" int va_validate_batch(va_ctx*, const void*, const void*, int, int, void*, void*) { va_set_error("not built yet"); return VA_ERR_INVALID; }
