// pt_scene_prep.h — device-side pieces of pt_scene_create (pt_scene_prep.hip), shared inside libpt_hip.so.
#pragma once
#include <stdint.h>

#include "../../include/pt_api.h"

namespace ptp {

// The tables that settle a tie on t in the caller's visit order (pt_trace.h: ref_visits_first), made on the device from the
// caller's node pool: for every primitive the turns of its root-to-leaf path (bit k = level k goes right) and the inner node
// of the re-laid caller's tree (inner_of_pool: pool index -> DNode index) at every level of that path.
// pool_dev / inner_of_pool_dev / path_dev [N] / anc_dev [N * levels]: device memory.  scratch_parent_dev: num_nodes int32.
// Every leaf walks up to the root; a leaf that does not get there within `levels` + 1 steps makes the call fail (the host has
// validated the tree: this cannot happen).
int tie_tables_device(const pt_bvh_node* pool_dev, int num_nodes, int root, const int32_t* inner_of_pool_dev, int N, int levels,
                      int32_t* scratch_parent_dev, unsigned long long* path_dev, int32_t* anc_dev);


// ---- a node pool (bvh.cuh:7-15: one node per leaf and per inner node, children by index) re-laid on the device -------------
// What pt_api.hip's convert_tree does on the host, for pools that are already in device memory: validation (caller's trees),
// the DFS pre-order of the inner nodes — for the library's own tree with the child that needs the shallower stack first
// (Strahler numbers) —, the most-visited top of the tree renumbered to the front, 64-B DNodes carrying both child boxes.
struct RelayResult {
    int32_t depth = 0;            // levels, leaves counting (computeMaxDepth, bvh.cu:56-65)
    int32_t stack_need = 0;       // entries a traversal can hold at once
    uint32_t top_avail = 0;       // nodes [0, top_avail) are the top of the tree in order of box area
    int32_t nested = 0;           // every node's box below the root contains its children's boxes
    int32_t num_inner = 0;
};
// pool_dev: num_nodes nodes, root its root; N primitives.  internal: the library's own tree (no validation; children ordered
// for a short stack, stack_need = Strahler number); otherwise the caller's (validated, PT_ERR_BAD_SCENE with the host path's
// messages; stack_need = depth - 1).  dnodes_dev: N - 1 DNode-sized records (64 B each).  inner_of_pool_dev: num_nodes.
// leaf_boxes_dev: optional N x 6 floats.  block_threads / top_nodes_max / lds_budget_max / node_bytes: what sizes the top prefix.
int relay_tree_device(const pt_bvh_node* pool_dev, int num_nodes, int root, int N, bool internal, void* dnodes_dev,
                      int32_t* inner_of_pool_dev, float* leaf_boxes_dev, int block_threads, uint32_t top_nodes_max,
                      uint32_t lds_budget_max, RelayResult* out);

// ---- primitive records on the device ---------------------------------------------------------------------------------------------
// One 48-B DPrim (+ one 48-B DNormals) per shape, gathered from the meshes (scene.h:179-184's pointer chase done once, shape.cuh:48-59's
// upload): the shapes and the mesh arrays go up as they are, one thread per shape writes the records.  Ids are checked on the
// device with the host path's messages (PT_ERR_BAD_SCENE).  *has_sphere: the scene holds a sphere.
int prims_device(const pt_scene_desc* d, void* prims_dev, void* normals_dev, int* has_sphere);

}  // namespace ptp
