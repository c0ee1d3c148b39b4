// pt_internal.h — shared by the translation units of libpt_hip.so (not part of the C ABI).
#pragma once
#include <string>

// Records the message pt_last_error() returns (thread-local) and hands `code` back.
int pt_fail(int code, const std::string& msg);

struct pt_bvh_node;
// pt_bvh_build.hip: SAH / LBVH hierarchy on the device over GIVEN primitive boxes (n x {min xyz, max xyz}, host memory);
// nodes come back in the reference layout, leaf boxes = the given boxes.  depth_cap > 0: no deeper than that (leaves count 1;
// at least ceil(log2 n) + 1), 0: the builder's default of ceil(log2 n) + 5.
int pt_bvh_build_from_boxes(const float* boxes, int n, int method, pt_bvh_node* out_nodes, int* out_root, int* out_depth,
                            int depth_cap);
