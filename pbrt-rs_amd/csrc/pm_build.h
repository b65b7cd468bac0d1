// Host-side scene storage and acceleration-structure build (internal to libpbrt_mi355.so).
#pragma once
#include <string>
#include <vector>

#include "pm_layout.h"

namespace pm {

struct Box { float lo[3], hi[3]; };

struct HostSphere {
    float radius, z_min, z_max, phi_max, theta_z_min, theta_z_max;
    float render_from_object[16], object_from_render[16];
    bool reverse_orientation;
    uint32_t prim_id;
    Box bounds;
};
struct HostPatch {
    float p[4][3];      // p00 p10 p01 p11
    float n[4][3];      // vertex normals (render space), valid when has_normals
    bool has_normals = false;
    float uv[4][2];     // vertex texture coordinates, valid when has_uv
    bool has_uv = false;
    bool flip_normal;
    uint32_t prim_id;
    Box bounds;
};
struct HostTri {
    float v[3][3];
    uint32_t prim_id;
    Box bounds;
};

struct HostScene {
    std::vector<HostSphere> spheres;
    std::vector<HostPatch> patches;
    std::vector<HostTri> tris;
    uint32_t next_prim_id = 0;
};

struct Accel {
    std::vector<PmWideNode> nodes;
    std::vector<uint8_t> prims;     // leaf records, 16-byte units
    Box bounds;
    double build_seconds = 0.0;
    // a tree built on the device (pm_gpu_build.hip) stays there: nodes / prims above are empty and these say how large the
    // device buffers are; pm_get_node_buffer / pm_get_prim_buffer fetch the host copies when somebody asks for them
    size_t dev_nodes = 0, dev_prim_bytes = 0;
    // flat table of a scene small enough to be swept without a tree (pm_layout.h PmFlatTri; empty otherwise):
    // flat_n[0] sphere record units, then flat_n[1] patch record units (uint32 each), in flat_units; flat_n[2] triangles
    std::vector<PmFlatTri> flat_tris;
    std::vector<uint32_t> flat_units;
    uint32_t flat_n[3] = {0, 0, 0};
};

int build_accel(const HostScene& sc, int max_leaf, Accel& acc, std::string& err);
// fills acc.flat_* from acc.nodes / acc.prims when the tree references at most max_prims primitives (clears them otherwise)
void build_flat_table(Accel& acc, uint32_t max_prims);

}  // namespace pm
