// pt_kernels.hpp -- the HIP kernels of libptmi355.so (included by ptmi355.hip only), counterparts of
// the reference's src/pathtrace.cu kernels: k_raygen (generateRayFromCamera :122-143), k_intersect
// (computeIntersections :149-213), k_bounce (intersect + shade/scatter + stable compaction, fused),
// k_iteration (all bounces of a small batch in one launch), k_sort_hist / k_shade_sorted_w / k_shade_sorted (material
// sort), k_mesh (triangle-mesh pre-pass), k_cull0_mask (bounce-0 candidate masks), k_shade_fake (shadeFakeMaterial
// :224-266), k_gather (finalGather :269-278), k_tonemap (sendImageToPBO :48-68) and the AoS import/export helpers.
#pragma once

// Split by kernel family (round 4); the order is the dependency order.
#include "pt_k_common.hpp"
#include "pt_k_scene.hpp"
#include "pt_k_bvh.hpp"
#include "pt_k_trisweep.hpp"
#include "pt_k_pool.hpp"
#include "pt_k_intersect.hpp"
#include "pt_k_sort.hpp"
#include "pt_k_bounce.hpp"
#include "pt_k_mesh.hpp"
#include "pt_k_image.hpp"
