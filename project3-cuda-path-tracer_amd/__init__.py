"""ptmi355 -- MI355X-native wavefront path tracer (host-side mirror of the
reference's renderer interface, src/pathtrace.h:6-8, over the C-ABI in
include/ptmi355.h).  The directory name carries a hyphen, so import it through
`__graft_entry__.load_package()` (alias `ptmi355`)."""
from .binding import (  # noqa: F401
    CAMERA_DT, GEOM_DT, ISECT_DT, MATERIAL_DT, MESH_DT, PATH_DT, TRI_DT,
    PT_AA_JITTER, PT_ASYNC_IMAGE, PT_PIN_IMAGE, PT_HOST_SPARSE, PT_SHARED_IMAGE, PT_LOOKAHEAD, PT_CACHE_FIRST, PT_COMPACT, PT_FAKE_SHADER, PT_MESH_BVH, PT_SORT_MATERIAL, PT_UNFUSED,
    PtError, Scene, Stats, library, pathtrace, pathtraceFree, pathtraceInit,
    device_image_ptr, export_intersections, export_paths, get_image, get_stats, intersect_once,
    set_camera, set_lens, synchronize, tonemap, trace_batch, trace_batch_async, trace_begin, trace_bounce,
    trace_end, clear_image, version, has_experiments, set_profiling, get_profile, STAGES, total_rays, counters, cull_boxes, num_devices, exchange_transport, tri_bounds, tri_records,
    set_image, probe_rng, probe_sincos, probe_sincos_sums, probe_hemisphere, probe_sqrt, probe_clock,
)
from .build import build  # noqa: F401
from . import sharding  # noqa: F401,E402
from . import meshes  # noqa: F401,E402
from . import host_binding  # noqa: F401,E402
from .host_binding import build_host, build_ptbench, image_to_rgb8, load_scene, save_pfm, save_png, load_pfm  # noqa: F401,E402
