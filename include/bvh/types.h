/* types.h — POD layouts of the BVH build path.  Binary-compatible with the reference's src/Common.h:
 *   Aabb      24 B  (src/Common.h:310-416)      Triangle  64 B, alignas(64), 36-B payload (src/Common.h:429-434)
 *   Bvh2Node  32 B, alignas(32) (src/Common.h:436-441)      PrimRef   28 B (src/Common.h:574-578)
 * Usable from C, C++ and HIP device code. */
#ifndef BVH_TYPES_H
#define BVH_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
#define BVH_ALIGNAS(n) alignas(n)
#else
#define BVH_ALIGNAS(n) _Alignas(n)
#endif

#define BVH_INVALID 0xFFFFFFFFu                 /* INVALID_NODE_IDX / INVALID_PRIM_IDX, src/Common.h:90-92 */
#define BVH_FLT_MAX 3.402823466e+38f            /* FltMax, src/Common.h:86 */

typedef struct { float x, y, z; } bvh_float3;
typedef struct { bvh_float3 min, max; } bvh_aabb;
typedef struct BVH_ALIGNAS(64) { bvh_float3 v1, v2, v3; } bvh_triangle;
typedef struct BVH_ALIGNAS(32) { uint32_t left, right; bvh_aabb aabb; } bvh2_node;
typedef struct { uint32_t prim_idx; bvh_aabb aabb; } bvh_primref;

#ifdef __cplusplus
static_assert(sizeof(bvh_aabb) == 24, "Aabb is 24 bytes");
static_assert(sizeof(bvh_triangle) == 64, "Triangle is 64 bytes");
static_assert(sizeof(bvh2_node) == 32, "Bvh2Node is 32 bytes");
static_assert(sizeof(bvh_primref) == 28, "PrimRef is 28 bytes");
#endif

#endif
