// mi355pt — host side of the "prefer fast trace" builder: binned-SAH topology over world-space triangle boxes (pt_build_sah.cpp); everything else of the build
// (bounds, leaves, BVH8 collapse, refit) is pt_build.hip's.
#pragma once
typedef unsigned int uint;
namespace ptk {
struct SahTri { float mn[3], mx[3], c[3]; };                        // box and centroid of one triangle (world space)
// n triangles: order[n] (depth-first leaf order of triangle ids), childL/childR/rangeFirst/rangeLast/parent [n - 1] (inner node 0 = root), leafParent[n]
// absorb[n - 1] (may be null): 1 = the cost-driven BVH8 collapse opens this inner node inside its parent's wide node (see bvh_sah_topology)
struct SahTopology { uint* order; uint* childL; uint* childR; uint* rangeFirst; uint* rangeLast; uint* parent; uint* leafParent; uint* absorb; };
// maxLeaf: sub-trees of at most this many triangles become one leaf (k_emit's rule). With out.absorb the wide-node assignment is chosen too: the dynamic
// programme of Ylitie, Karras & Laine 2017 (every wide-node visit and every leaf visit costs its surface area) instead of "open the largest child".
// Returns the number of re-insertion passes that ran.
uint bvh_sah_topology(const SahTri* tris, uint n, const SahTopology& out, uint maxLeaf, unsigned threads /* 0 = the hardware's threads, at most 32 */);
}
