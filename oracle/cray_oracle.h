/*
 * cray_oracle.h — CPU restatement of the c-ray hot path (TEST INFRASTRUCTURE, see cray_oracle.c).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#ifndef CRAY_ORACLE_H
#define CRAY_ORACLE_H
#include "cray_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* renderThread() restated: passes [first_pass, first_pass+pass_count) of every pixel of the region,
 * folded into fb (host, W*H*3 floats, the reference's y-flipped layout). threads <= 0: all cores. */
int oracle_render_region(const crh_scene_desc *scene, const crh_render_params *params, float *fb,
						 crh_counters *counters_out, int threads);

/* getClosestIsect() restated for caller-supplied world-space rays (6 floats each). */
int oracle_trace_rays(const crh_scene_desc *scene, const float *rays, uint64_t n, crh_hit *hits);

/* colorToSRGB + setPixel 8-bit truncation (color.h:60-84, texture.c:18-22). */
void oracle_to_srgb8(const float *fb, int width, int height, uint8_t *rgb8);

/* initSampler(Random) + n getDimension() draws for (pixel, pass) — sampler.c:41-44, random.c:16-21. */
void oracle_sampler_draws(uint32_t pixel_index, int pass, int max_passes, int n, float *out);

/* getCameraRay() for pixel (x, y) and (pass, max_passes): 6 floats out (start, direction). */
void oracle_camera_ray(const crh_scene_desc *scene, int x, int y, int pass, int max_passes, float *out6);

#ifdef __cplusplus
}
#endif
#endif
