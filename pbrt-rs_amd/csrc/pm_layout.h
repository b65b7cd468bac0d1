// Device data layout of the wide quantised BVH and the leaf primitive records (HBM-resident).
// Replaces the reference's 64-byte binary LinearBVHNode (primitives/aggregates.rs:719-732) and the
// Vec<Arc<PrimitiveEnum>> -> Arc<ShapeEnum> -> mesh pointer chase (primitives/single.rs:72-75,
// shapes/bilinear_patch.rs:700-707).
#pragma once
#include <stdint.h>

// 80-byte 8-ary node.  Child boxes are stored as 8-bit offsets on a per-node power-of-two grid:
//   plane = fmaf((float)q, 2^(e-127), o)      (single rounding; the builder rounds q outward
//                                               against exactly this expression)
// Children that are interior nodes (bit set in imask) live at child_base + rank(slot in imask).
// Leaf children reference `count` consecutive records starting `offset` records after prim_base.
// All leaf children of one node hold the same primitive type (type-homogeneous subtrees).
struct alignas(16) PmWideNode {
    float o[3];
    uint8_t e[3];
    uint8_t imask;
    uint32_t child_base;
    uint32_t prim_base;      // bits 0..29: first record, in 16-byte units; bits 30..31: PM_PRIM_* type
    uint8_t meta[8];         // 0 = empty slot or interior; leaf: (count << 5) | offset (records)
    uint8_t qlo[3][8];
    uint8_t qhi[3][8];
};
static_assert(sizeof(PmWideNode) == 80, "node must be 80 bytes");

enum { PM_PRIM_SPHERE = 0, PM_PRIM_PATCH = 1, PM_PRIM_TRI = 2 };

// record sizes in 16-byte units
#define PM_UNITS_SPHERE 5
#define PM_UNITS_PATCH 4
#ifndef PM_UNITS_TRI
#define PM_UNITS_TRI 3
#endif

struct alignas(16) PmTriRecord {      // 48 B: degenerate patch p00=v0 p10=v1 p01=v2 p11=v2
    float v0[3]; float v1[3]; float v2[3];
    uint32_t prim_id; uint32_t pad[2 + 4 * (PM_UNITS_TRI - 3)];
};
struct alignas(16) PmPatchRecord {    // 64 B
    float p00[3]; float p10[3]; float p01[3]; float p11[3];
    uint32_t prim_id; uint32_t flags; uint32_t pad[2];
};
struct alignas(16) PmSphereRecord {   // 80 B; object_from_render rows 0..2 (affine: row 3 == 0,0,0,1)
    float ofr[12];
    float radius, z_min, z_max, phi_max;
    uint32_t prim_id; uint32_t flags;
    float m_norm;     // max over rows of |m_r0| + |m_r1| + |m_r2|   } error scales of the float pre-filter
    float t_norm;     // max over rows of |m_r3|                     } (pm_device.hpp sphere_prefilter)
};
static_assert(sizeof(PmTriRecord) == 16 * PM_UNITS_TRI && sizeof(PmPatchRecord) == 64 && sizeof(PmSphereRecord) == 80, "record sizes");

// Scenes of at most PM_FLAT_MAX_PRIMS primitives are not walked at all: every wave sweeps ALL primitives in lock step
// (wave-uniform records, no divergence, no node steps), which beats any tree of that size on 64-lane SIMT hardware.
// One 64-byte entry per triangle, derived from its record: e1 = v1 - v0, e2 = v2 - v0 (f32), n = e1 x e2 (f64 products,
// rounded once), L = max |component| of e1, e2; c1 = max |component| of v0 + L and kll = (K * L) * L (K = 128 * 2^-24)
// are the triangle's share of the pre-filter's error bounds; `unit` locates the 48-byte record for the exact test.
// Table ORDER (pm_build.cpp build_flat_table): entries with bit-equal v0 are adjacent - a quad's two triangles, the faces
// of a box around a corner - and inside such a group an entry whose e1 is the previous entry's e2 (the quad's diagonal)
// follows it.  The top bits of `unit` say so, and the sweep then keeps what depends on v0 alone (ro - v0 and its cross
// product with the direction: 9 operations) and the shared edge's triple product (3) from the previous entry.  Same
// operands, same operations: the values are those of a fresh evaluation, bit for bit.
#define PM_FLAT_SAME_V0 0x40000000u     /* v0 == the previous entry's v0 */
#define PM_FLAT_SAME_EDGE 0x80000000u   /* ... and e1 == the previous entry's e2 */
#define PM_FLAT_SAME_N 0x20000000u      /* ... and n == the previous entry's n (a flat quad): rd.n, the facing and sv.n are kept too */
#define PM_FLAT_UNIT_MASK 0x1FFFFFFFu
#define PM_FLAT_MAX_PRIMS 64
struct alignas(16) PmFlatTri {
    float v0[3]; float L;
    float e1[3]; uint32_t unit;
    float e2[3]; float c1;
    float n[3];  float kll;
};
static_assert(sizeof(PmFlatTri) == 64, "flat triangle entry is 64 bytes");
// kernel argument: device pointers; units = n_sphere sphere record units, then n_patch patch record units
struct PmFlatScene { const PmFlatTri* tris; const uint32_t* units; uint32_t n_sphere, n_patch, n_tri; };

#define PM_MAX_LEAF 4
#ifndef PM_LDS_STACK
#define PM_LDS_STACK 6       // per-lane entries kept in LDS (6 KB per 128-lane block: with the candidate queue, the ring of
                             // parked leaves and the permutation table a block takes 14 KB, so the 10 blocks = 5 waves per
                             // SIMD that the kernels' registers allow fit the CU's 160 KB.  Measured with the ring in place
                             // (profiles/r02/ab_deferl.log): 8 entries = 16 KB 57 ms, 6 = 54 ms, 5 = 56 ms, 4 = 59 ms on the
                             // C4-small closest-hit launches; 12 entries held the walk to 4 waves per SIMD)
#endif
#ifndef PM_STACK_TOTAL
#define PM_STACK_TOTAL 48    // traversal stack entries per ray; one entry per tree level at most (a group of hit siblings still to
                             // visit).  A wide level is at least one binary level, so the device LBVH (63-bit Morton codes: binary
                             // depth <= 63 + log2 n) and any SAH tree of a sane scene stay far below it.  Beyond: the group is
                             // dropped and the call returns PM_ERR_STACK_OVERFLOW (tests/test_stack_overflow_gpu.py provokes it
                             // with the 3-entry build `make smallstack`).
#endif
#define PM_SPILL_STACK (PM_STACK_TOTAL - PM_LDS_STACK)    // per-lane overflow entries (private memory)
