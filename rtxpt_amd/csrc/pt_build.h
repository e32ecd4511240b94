// mi355pt — GPU LBVH build / refit (replaces the driver's BLAS/TLAS build: Rtxpt/Sample.cpp:1061-1079, 1170-1240).
//
//   k_tri_setup     one thread per triangle: instance transform -> world-space TriRecord + scene bounds (wave-reduced atomics)
//   k_morton        63-bit Morton code of the centroid (21 bits / axis)
//   rocprim sort    (code, primitive) pairs — library radix sort, build step only
//   k_ploc_*        (default) PLOC, Meister & Bittner 2018: mutual-nearest-neighbour merging of the Morton-ordered clusters inside a window of
//                   PT_PLOC_RADIUS, ~70 data-parallel passes; leaves are then renumbered in depth-first order so that every node covers a
//                   contiguous leaf range again. Half the SAH cost of the Karras tree on C3 (tools/bvh_lab), 40 % fewer node visits per ray
//   pt_build_sah    (BVH_BUILDER_SAH, round 2's "prefer fast trace" builder, now behind PT_DEVICE_HOST_SAH_BUILDER) binned-SAH topology built on the host's
//                   cores from the world-space triangle boxes; bounds, leaves, collapse and refit below are shared. 14-18 % fewer node visits than PLOC
//   k_ri_*          (BVH_BUILDER_PLOC_OPT, the default: "prefer fast trace" on the device) parallel re-insertion on the finished PLOC tree (pt_build_reinsert.h): per pass search /
//                   lock / check / ring test / apply + a level-ordered refit of the whole tree; twelve passes, 4.4 ms each at 2.8 M triangles
//   k_wide_*        the cost-driven wide-node assignment (pt_build_wide.h) over the levels of a breadth-first numbering: which inner nodes a BVH8 node opens
//   k_karras        (PT_BVH_BUILDER=karras) Karras 2012 hierarchy over the sorted codes (duplicate codes split on the index bits)
//   k_leaf_boxes / k_range_level / k_node_boxes
//                   node bounds WITHOUT inter-thread hand-offs: every Karras node covers a contiguous range of sorted leaves, so its two child
//                   boxes are two range-min/max queries on a sparse table over the leaf boxes (log2 n fully parallel passes, 2 GB for 2.8 M
//                   triangles — HBM is 288 GB). The classic bottom-up pass with an atomic ticket and agent-scope fences per node (k_bounds, kept
//                   for scenes above PT_RANGE_TABLE_MAX_TRIS) took 20.9 ms of a 25 ms refit; min/max are exact, so the tree is bit-identical
//   k_emit          collapse sub-trees of <= 4 triangles into leaves and write BVH2 nodes (both child boxes per node)
//   k_alpha_records per leaf-order triangle: texture coordinates + alpha texture + cutoff for the traversal's alpha test
//   k_collapse8     level by level: greedily open the largest-area inner child until 8 children -> quantised 128 B BVH8 nodes
// Refit (animated instances / deformed vertices, same topology), round 4: k_refit_world (leaf-order triangles straight from a flat source record: instance, three global
//   vertex indices — one hop instead of five; scene bounds reduced per 1024-thread block before the six atomics) and k_refit8_level, bottom-up over the levels of the WIDE
//   tree: a wide node's children keep their slots, their boxes are the padded unions of what lies below — triangles of a leaf child, the stored un-padded box of an inner
//   child — quantised by the expression the collapse uses (bvh8_pack). min / max are exact, so the nodes are the bytes a fresh bounds + emit + collapse over the same
//   topology writes, without the 21-level sparse table, the BVH2 nodes and the level-by-level host round trips of the collapse: 5.2 -> 0.5 ms at 2.8 M triangles.
#pragma once
#include "pt_scene.h"
#include <hip/hip_runtime.h>

namespace ptk {

enum : uint { BVH_BUILDER_PLOC = 0, BVH_BUILDER_KARRAS = 1, BVH_BUILDER_SAH = 2, BVH_BUILDER_PLOC_OPT = 3 };      // SAH: topology by pt_build_sah.cpp on the host; PLOC_OPT: PLOC + re-insertion passes + cost-driven wide nodes, all on the device (both "prefer fast trace")

struct TriSrc { uint instance, i0, i1, i2, prim, flags, _pad0, _pad1; };      // leaf order: what k_tri_setup gathers through primInfo -> subInstToInstGeom -> {instance, geometry, sub-instance} -> indices, resolved once per build
static const uint BVH_MAX_WIDE_LEVELS = 64;

struct BvhBuildBuffers {
    TriRecord* triWorld;        // by global primitive id
    TriRecord* triSorted;       // leaf order
    unsigned long long* keys; unsigned long long* keysSorted;
    uint* prims; uint* primsSorted;
    uint* childL; uint* childR; uint* parent; uint* leafParent; uint* rangeFirst; uint* rangeLast;
    uint* tickets;
    float4* boxLmin; float4* boxLmax; float4* boxRmin; float4* boxRmax;
    uint* sceneBounds;          // 6 ordered-uint encoded floats (min xyz, max xyz)
    BvhNode* nodes;
    AlphaRec* alphaRecs;        // leaf order, parallel to triSorted
    float4* rangeMin; float4* rangeMax; uint rangeLevels;   // sparse table over the leaf-order triangle boxes: level k, entry i = bounds of leaves [i, i + 2^k)
    uint* primToSlot;           // global primitive id -> leaf-order slot (k_resolve_extend looks the winning triangle up by primitive)
    Bvh8Node* nodes8; uint* levelA; uint* levelB; uint* wideCounter; uint numNodes8, collapseLevels;
    TriSrc* triSrc;             // leaf order (refit)
    float4* wideBoxMin; float4* wideBoxMax;      // per wide node: the un-padded box of everything below it (refit: an inner child's box in its parent)
    uint wideLevelStart[BVH_MAX_WIDE_LEVELS + 1]; uint wideRefitReady;
    bool bvh2Stale = false;      // set by the fast refit (k_refit_world + k_refit8_level): triWorld, the BVH2 `nodes` and the range tables still hold the pose of the last full build — anything that reads them
                                 // (tools, a BVH2 probe, re-insertion) must rebuild first; the traversal reads triSorted / nodes8 only. pt_api.hip publishes dsc.nodes = null while this is set.
         // wide nodes of collapse level L are [wideLevelStart[L], wideLevelStart[L + 1]): children always lie in a deeper level
    void* sortTemp; size_t sortTempBytes;
    // PLOC work arrays (node ids while building: leaves 0..n-1 in Morton order, inner nodes n..2n-2 in creation order)
    uint* plocCl[2]; uint* plocNN; unsigned long long* plocFlags; unsigned long long* plocOffs; uint* plocChildA; uint* plocChildB; uint* plocCnt; uint* plocParent; uint* plocFirst; uint* plocCounts;
    void* scanTemp; size_t scanTempBytes; uint plocPasses;
    uint builder;               // BVH_BUILDER_PLOC ("prefer fast build"), BVH_BUILDER_SAH ("prefer fast trace") or BVH_BUILDER_KARRAS (developer A/B)
    uint* absorb;               // BVH_BUILDER_SAH: per inner node, 1 = the cost-driven BVH8 collapse opens it inside its parent's wide node (pt_build_sah.cpp)
    float hostBuildMs;          // BVH_BUILDER_SAH: the host part of the last build (read-back + SAH topology + upload)
    void* riScratch; size_t riScratchBytes; uint riPasses, riLevels;      // BVH_BUILDER_PLOC_OPT: scratch of the parallel re-insertion, passes to run (pt_api: 10), depth of the tree it left
    uint wideDpPending;         // BVH_BUILDER_PLOC_OPT: the next bounds stage runs the device-side wide-node programme (a build; refits keep the flags they find)
    uint wideFlagsValid;        // BVH_BUILDER_PLOC_OPT: absorb[] holds the programme's flags (0: the collapse opens the largest child)
    uint wideLevels;            // depth of the binary tree above the wide tree's leaves (levels of the last device-side wide-node programme)
    uint optimiserPasses;       // re-insertion passes the last build ran (host: pt_build_sah.cpp; device: BVH_BUILDER_PLOC_OPT)
    uint capacity;
};

hipError_t bvh_alloc(BvhBuildBuffers& b, uint numTris);
void bvh_free(BvhBuildBuffers& b);
// full build: fills b.triSorted / b.nodes; scene must already reference primInfo/instances/geometries/streams
hipError_t bvh_build(BvhBuildBuffers& b, const DeviceScene& sc, uint numTris, hipStream_t stream);
// flat per-primitive shading records (pt_scene.h ShadeTri) of primitives [firstPrim, firstPrim + numTris); rebuilt when the geometry is (re)set or its vertices are deformed
void launch_shade_tris(const DeviceScene& sc, uint firstPrim, uint numTris, ShadeTri* out, hipStream_t stream);
hipError_t bvh_refit(BvhBuildBuffers& b, const DeviceScene& sc, uint numTris, hipStream_t stream);

} // namespace ptk
