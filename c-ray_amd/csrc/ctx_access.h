/* ctx_access.h — what the other translation units of libcray_hip.so may know about a crh_ctx (hidden symbols). */
#pragma once
#include "cray_hip.h"
extern "C" {
__attribute__((visibility("hidden"))) int crh_internal_device(crh_ctx *ctx);
__attribute__((visibility("hidden"))) void *crh_internal_stream(crh_ctx *ctx);
__attribute__((visibility("hidden"))) void *crh_internal_pinned(crh_ctx *ctx, size_t bytes);   /* the context's page-locked host scratch, grown to at least `bytes` (contents lost when it grows); NULL when it cannot */
__attribute__((visibility("hidden"))) int crh_internal_fail(int code, const char *message);   /* sets crh_last_error(), returns code */
}
