/* flatten.h — struct world -> crh_scene_desc (see flatten.c). Host-side C, reference-facing. */
#pragma once
#include "cray_hip.h"

struct renderer;

/* Fills `out` with freshly malloc'ed arrays describing r->scene; release with crh_flatten_free(). */
int  crh_flatten_world(const struct renderer *r, crh_scene_desc *out);
void crh_flatten_free(crh_scene_desc *d);
/* The struct prefs fields (renderer.h:58-87) the hot path needs. */
crh_blob_prefs crh_flatten_prefs(const struct renderer *r);
