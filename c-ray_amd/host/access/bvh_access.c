/* Wrapper TU: compiles the UNMODIFIED reference src/accelerators/bvh.c (builder + traversal) and
 * appends read-only accessors for the file-private `struct bvh` (bvh.c:44-48). See describe.h. */
#include "accelerators/bvh.c"
#include "describe.h"

const void *crh_access_bvh_nodes(const struct bvh *b) { return b ? b->nodes : NULL; }
const int  *crh_access_bvh_prims(const struct bvh *b) { return b ? b->primIndices : NULL; }
unsigned    crh_access_bvh_node_count(const struct bvh *b) { return b ? b->nodeCount : 0; }

_Static_assert(sizeof(struct bvhNode) == sizeof(crh_bvh_node), "crh_bvh_node must mirror struct bvhNode (32 B)");
