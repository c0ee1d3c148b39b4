// pt_sweep_build.h — device build of the library's internal tree (pt_sweep_build.hip), shared inside libpt_hip.so.
#pragma once
#include <stdint.h>

#include <vector>

#include "../../include/pt_api.h"

namespace pts {

// The tree pts::build_sweep_tree (pt_tree_sweep.h) makes of the same leaf boxes, byte for byte, built on the current HIP device.
// leaf_boxes: n x {lo.xyz, hi.xyz} on the HOST, all finite.  out: 2n-1 nodes in pre-order, root 0 (host memory).
// Returns PT_OK; PT_ERR_UNSUPPORTED when the input drives a branch past the host builder's depth guard (the caller then runs
// the host builder: the median-cut fallback lives there only); PT_ERR_DEVICE / PT_ERR_NO_DEVICE on HIP errors.
int sweep_build_device(const float* leaf_boxes, int n, std::vector<pt_bvh_node>& out, int32_t* out_root, int32_t* out_depth,
                       double* out_device_ms);

// The same with both ends on the device: leaf_boxes_dev n x 6 floats, nodes_dev 2n-1 nodes (pre-order, root 0).
int sweep_build_on_device(const float* leaf_boxes_dev, int n, pt_bvh_node* nodes_dev, int32_t* out_depth, double* out_device_ms);

}  // namespace pts
