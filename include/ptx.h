/*
 * ptx.h -- C ABI of libptx_hip.so: the MI355X (gfx950) progressive path tracer.
 *
 * This is the drop-in boundary for the hot path of aminequirogabritos/path-tracing.  The reference has
 * no FFI; the seam it replaces is the implicit WebGL contract between main.js and
 * shaders/fragmentShaderPathTracing.glsl.  Each entry point cites the reference call
 * site it stands in for (paths relative to /root/reference).  Plain pointers and sizes only; the caller
 * keeps ownership of everything it passes in; nothing here retains host pointers after returning.
 *
 * Conventions: functions return PTX_OK (0) or a negative ptx_status; the message of the last failure is
 * available from ptx_last_error().  No exception crosses the ABI.  A context is not thread-safe (the
 * reference host is single-threaded JS, main.js:262,391).  Multi-GPU (DESIGN.md "Multi-GPU"): frames are
 * split over the GPUs and the float accumulation buffers are summed by ONE all-reduce (RCCL over xGMI, loaded
 * on first use) -- either inside one context created over several devices (ptx_create(device_ids, n > 1): the
 * single-process form the Node host uses), or between one single-device context per process
 * (ptx_comm_unique_id / ptx_comm_init / ptx_allreduce).
 * There is no CPU fallback: every entry point fails with PTX_ERR_HIP when no gfx950 device is usable.
 */
#ifndef PTX_H
#define PTX_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ptx_ctx ptx_ctx;

typedef enum ptx_status {
    PTX_OK = 0,
    PTX_ERR_INVALID = -1, /* bad argument (NULL, size mismatch, index out of range) */
    PTX_ERR_HIP = -2,     /* a HIP runtime call failed; message in ptx_last_error   */
    PTX_ERR_STATE = -3,   /* call order: scene / params not set yet                 */
    PTX_ERR_NOMEM = -4
} ptx_status;

/*
 * The eleven data textures main.js:272-282 uploads with utils/textureUploader.js:4 (coordinatesTexture,
 * normalsTexture, colorsTexture, emissionsTexture, lightIndicesTexture, nodesBoundingBoxesMins,
 * nodesBoundingBoxesMaxs, nodesMissLinkIndices, nodesInorderTrianglesIndices, nodesTrianglesCount,
 * inorderTrianglesIndicesArray), de-triplicated and typed (the reference stores every integer as a
 * float copied into r,g,b: classes/bvh.js:284-293, main.js:187-190, utils/triangleMapper.js:62-64).
 */
typedef struct ptx_scene {
    uint32_t T; /* triangles       (uniform triangleCount,     main.js:298) */
    uint32_t N; /* BVH nodes       (uniform bvhNodeCount,      main.js:305) */
    uint32_t L; /* light triangles (uniform lightIndicesCount, main.js:299) */
    const float *coords;        /* 9T: a.xyz b.xyz c.xyz per triangle, world space            */
    const float *normals;       /* 3T: geometric normal per triangle (main.js:652-653)        */
    const float *colors;        /* 3T: material.color                                          */
    const float *emissions;     /* 3T: material.emissive * 50 (utils/triangleMapper.js:44,60)  */
    const int32_t *light_tri;   /* L : indices of triangles with any emissive channel > 0      */
    const float *node_min;      /* 3N: AABB min per node, pre-order (classes/bvh.js:282)       */
    const float *node_max;      /* 3N: AABB max per node                                       */
    const int32_t *node_miss;   /* N : miss link = pre-order index after the subtree, -1 = end */
    const int32_t *node_first;  /* N : leaf: first slot in inorder_tri; internal: -2           */
    const int32_t *node_count;  /* N : leaf: triangle count; internal: 0                       */
    const int32_t *inorder_tri; /* T : triangle indices, leaf lists concatenated in pre-order  */
} ptx_scene;

/*
 * The gl.uniform* block of main.js:290-308 that the shader actually reads.  aspectRatio is derived as
 * the reference does (JS double width/height rounded to f32, main.js:291).  Not needed: cameraDirection,
 * vertexCount, totalFrames, quadX, quadY, quadSize, maxTextureSize (addressing / tiling only).
 */
typedef struct ptx_params {
    uint32_t width, height;   /* uniform windowSize */
    float cam_src[3];         /* cameraSource       */
    float cam_up[3];          /* cameraUp           */
    float cam_right[3];       /* cameraRight        */
    float cam_left_bottom[3]; /* cameraLeftBottom   */
    int32_t max_path_length;  /* maxPathLength      */
    int32_t sample_count;     /* sampleCount (paths per pixel per frame) */
} ptx_params;

/* Work counters since the last ptx_reset_counters (new; for Mrays/s and the roofline). */
typedef struct ptx_counters {
    uint64_t samples;        /* camera paths (get_ray_radiance calls, glsl:404)                          */
    uint64_t rays;           /* closest-hit queries the REFERENCE algorithm issues for those samples     */
    uint64_t rays_traced;    /* traversals actually executed (primary-hit cache + shadow-ray memo remove some) */
    uint64_t node_visits;    /* box tests actually executed   (only with option "count" = 1, else 0)     */
    uint64_t tri_tests;      /* triangle tests actually executed (only with option "count" = 1)          */
    uint64_t ray_hits;       /* reference-equivalent rays that hit something                             */
    uint64_t light_attempts; /* iterations of sample_random_light's loop (glsl:372)                      */
    uint64_t pixel_frames;   /* (pixel, frame) results accumulated                                       */
    uint64_t kernel_launches;/* launches of the path-tracing kernel                                      */
    double kernel_ms;        /* HIP-event time of those launches, summed (valid after ptx_synchronize).  With option
                                "overlap" successive calls run on two streams and their spans overlap: the sum then
                                exceeds the wall time of a loop of small calls (one call = one span: exact)        */
    uint64_t diag[12];       /* SIMT-efficiency diagnostics, only with option "count" = 1: lane-iterations that did work and
                                64 x wave-iterations for {node walk, triangle tests, traversal calls}: useful, slots pairs; then shader
                                cycles and wave-iterations per phase {walk, tri, shade} of pipelines 1 and 2 */
} ptx_counters;

/* ---- lifecycle (no reference counterpart: canvas.getContext('webgl2'), main.js:123) ---- */
int ptx_device_count(void);
/* device_ids: HIP device ordinals, n_devices in 1 .. 64.  n_devices == 1: a context bound to that GPU.  n_devices > 1:
 * one context over several GPUs -- it replaces the single WebGL context of main.js:123 for a whole node: the scene and the
 * parameters go to every device, ptx_render_frames deals the frames of each call round-robin (frame f of the call to device
 * f mod n, over the full image), the devices render concurrently, and the first read-back after rendering sums the
 * per-device accumulation buffers with one grouped ncclAllReduce (communicators from ncclCommInitAll).  The ids must be
 * all distinct; the same id n times is accepted as a one-GPU rehearsal of the same code path (the sum is then taken by a
 * kernel in device order, RCCL cannot put two ranks on one device).  Not available on a multi-device context: option
 * "compat_rgba8" / ptx_read_compat_rgba8 (an 8-bit blend that is sequential over frames) and ptx_comm_init / ptx_allreduce.
 * The float sum of n devices differs from a one-device render of the same frames only by the association of the
 * additions.  Returns NULL on failure (see ptx_last_error(NULL)). */
ptx_ctx *ptx_create(const int *device_ids, int n_devices);
/* width and height of the last ptx_set_params (PTX_ERR_STATE before it): the size of every read-back buffer. */
int ptx_get_size(const ptx_ctx *ctx, uint32_t *width, uint32_t *height);
void ptx_destroy(ptx_ctx *ctx);
const char *ptx_last_error(const ptx_ctx *ctx); /* ctx may be NULL: last creation error */

/* ---- scene + parameters ---- */
/* Replaces the 11 uploadTexture() calls per frame of main.js:272-282 (done once here). Validates every
 * index (miss links, leaf ranges, triangle and light indices) before anything reaches the GPU. */
int ptx_upload_scene(ptx_ctx *ctx, const ptx_scene *scene);
/* Replaces the uniform block of main.js:290-308. Resets the accumulation when the size changes. */
int ptx_set_params(ptx_ctx *ctx, const ptx_params *params);
/* Tuning / test switches (no reference counterpart). Keys:
 *   "traversal" : bit 0 = best-t pruning (nodes that cannot hold a closer hit are skipped; the rule is option "prune"),
 *                 bit 1 = walk the leaf-SAH tree (same leaves, same results, about half the box tests; DESIGN.md
 *                 "Leaf-SAH tree").  0 = literal threaded traversal of glsl:207-249 on the reference tree, 1 = + pruning,
 *                 2 = leaf-SAH tree, 3 = both (default)
 *   "prune"     : the pruning rule of traversal bit 0 for BOUNCE rays (primary rays are always traced unpruned, so the
 *                 primary hit of every pixel is the reference's closest hit unconditionally).
 *                 2 (default) = margin rule: a node is skipped when its box is entered beyond
 *                 1.001 * best + 0.001 + lat, lat = margin_lat x scene diagonal x max|1/d_k| of the ray (below).  Faster, NOT
 *                 proven.  What it guards against: the shader's triangle test is ill-conditioned for a hit that grazes a
 *                 triangle (determinant near its 1e-4 cut-off); it then accepts points beside the triangle, whose leaf box
 *                 the ray enters later than the computed t -- by that distance times max|1/d_k| -- and if another hit lies in
 *                 between, the pruned walk returns that other hit.  Round 2's form of the rule (lat = 0) did get six pixels
 *                 of C5's whole job wrong that way (256 frames of 3840 x 2160, 726 G rays; tools/whole_job_check.py), and
 *                 ptx_selfcheck_pruning's adversarial rays break it ~20 times per 1e7.  With the default lat every frame of
 *                 every BASELINE job equals the unpruned walk bit for bit (profiles/r03_whole_job_check.json) and the
 *                 adversarial generator finds nothing in 1e8 rays per BASELINE scene (in 1e10: 8 rays on teapot, fired at a
 *                 grazing angle of 2^-24 rad from hundreds of units outside the scene; none on the other scenes).
 *                 1 = provable rule: a node is skipped only when a rounding-error bound of the shader's own triangle test
 *                 shows that no triangle beneath it can yield a computed t below the best one (DESIGN.md "Pruning and
 *                 exactness"): identical to the unpruned reference for EVERY ray, but the bound is so wide for triangles
 *                 that are large against the shader's absolute 1e-4 determinant cut-off that little is pruned there (pixar_room:
 *                 1.66 x the time of rule 2; the other BASELINE scenes: 1.11 - 1.18 x).  Callers that need the guarantee use it or
 *                 traversal = 2.
 *   "margin_lat": the margin rule's lateral allowance in ppm of the scene's diagonal (0 .. 1000000; default 3000, i.e. 300 x
 *                 the largest distance beside a triangle seen in the failures above; 0 = round 2's rule).  Headline cost of
 *                 the default against 0: 58.7 -> 59.7 ms per frame.  Ignored by "prune" = 1.
 *   "anyhit"    : 1 (default) = light-visibility rays (is_light_visible, glsl:353-363, and the shadow ray of glsl:439-442)
 *                 stop at the first conclusive blocker instead of finding the closest hit: the shader uses only "is there a
 *                 hit" and the closest hit's emission, and a hit on a triangle of emission (0,0,0) that no emissive triangle
 *                 can precede -- decided by the rule of "prune" on boxes around the emissive triangles -- fixes both.  Same
 *                 images and reference ray counters; pipeline 2 with pruning only.  0 = every ray is a closest-hit query
 *   "emit_list" : 1 (default) = under "prune" = 1, scenes with at most 32 emissive triangles decide "no emitter can be hit
 *                 nearer" with a bound on each emitter's determinant for the ray at hand instead of the shader's 1e-4 cut-off
 *                 (which on pixar_room switches the any-hit rule off: 133 -> 106 ms per headline frame, 98 with round 3's tighter M); 0 = round 2's bound
 *   "memo"      : 1 = reuse the per-pixel primary hit and the last light-attempt ray for the shadow ray
 *                 (default; both are bit-identical rays in the reference), 0 = trace every ray
 *   "count"     : 1 = also count node visits / triangle tests (slower), 0 = off (default)
 *   "pipeline"  : 0 = one lane per (pixel, frame), rays traced one at a time; 1 = wave-scheduled, per-lane leaf queues:
 *                 per-lane walk / triangle / shade work voted per wave, persistent lanes drawing tasks (ptx_wave.h);
 *                 2 = wave-scheduled with one leaf queue per wave shared by its lanes (ptx_pool.h; default; scenes whose
 *                 leaves share inorder slots are rendered by pipeline 1 instead)
 *   "compat_rgba8": 1 = also keep the reference's own progressive result: every frame is blended into an RGBA8 target
 *                 exactly as glsl:532-552 + modules/bufferManager.js:18 do (8-bit rounding per frame, the aspect-shifted
 *                 read of the previous target included); read it with ptx_read_compat_rgba8.  0 = off (default).
 *                 GL leaves the target's float -> 8-bit store and the texel chosen on an exact texel boundary to the
 *                 implementation: 1 uses the preferred rules (round to nearest; floor), 2 those of SwiftShader, the
 *                 implementation behind tests/golden/ref_frames, whose frames it reproduces byte for byte
 *   "overlap"   : 1 (default) = successive ptx_render_frames calls alternate between two launch slots (stream, frame buffers,
 *                 task counter each), so the kernel of call k + 1 fills the chip while the last, longest paths of call k
 *                 finish; the frame-order sum is kept by events.  A loop that renders one frame per call without
 *                 synchronising in between (main.js:400-405) then runs at the batched rate.  0 = one launch after the other
 *   "frame_batch": max frames rendered concurrently (0 = auto)
 *   "tasks_per_wave": (pixel, frame) tasks one wave of pipeline 1 works through (default 128)
 *   "pool_chunk": pipeline 2 runs one chip residency of persistent waves; each draws this many tasks at a time from a
 *                 global counter until the launch is exhausted (default 64 = one 8x8 tile)
 */
int ptx_set_option(ptx_ctx *ctx, const char *key, int64_t value);

/* ---- rendering ---- */
/* Replaces the per-frame draw of main.js:330-355 (gl.drawArrays per 32x32 tile + sleeps) for n_frames
 * frames: frame f runs sample_count paths per pixel seeded with timestamps[f] (uniform timestamp,
 * main.js:300; glsl:520) and adds clamp(mean, 0, 1) (glsl:526-529) to the float accumulation buffer, in
 * frame order.  first_frame_number mirrors uniform frameNumber (main.js:303) and is only book-keeping
 * for the float accumulation: the running mean sum/frames_done equals the reference's mix(prev, cur, 1/(frameNumber+1))
 * chain (glsl:537-540) without its RGBA8 rounding; with option "compat_rgba8" it is the frameNumber of that chain.  Asynchronous: work is queued on the context's stream. */
int ptx_render_frames(ptx_ctx *ctx, const int32_t *timestamps, uint32_t n_frames, uint32_t first_frame_number);
/* The same rendering for callers that cannot allocate.  No counterpart in the reference: it is for tight serving loops and
 * harnesses that forbid hipMalloc.  Same arguments, same kernels and bit-identical results
 * as ptx_render_frames, but every intermediate (the frame buffers of a batch, its timestamps, the task counter) lives in a
 * file-scope static __device__ array of 64 MiB + 512 B per GPU (held until the process ends) instead of library
 * allocations: the call makes no hipMalloc / hipFree / hipEventCreate and no host-to-device copy.  It is NOT capturable
 * into a HIP graph: the work goes to the library's own streams, and the first call per camera (or after a change of option
 * "traversal" / "prune") fills the per-pixel primary-hit cache and waits for it (one hipStreamSynchronize; the first use
 * of a kernel variant also runs an occupancy query).  Limit: width * height
 * <= 4 194 304 pixels (64 MiB / 16 B; frames are batched to fit), else PTX_ERR_INVALID with a message.  Scene and params
 * must already be set (ptx_upload_scene / ptx_set_params allocate; this call does not); option "compat_rgba8" only if an
 * earlier ptx_render_frames created its 8-bit targets.  The launches are not event-timed (ptx_counters.kernel_ms does
 * not include them).  All contexts on one GPU share that GPU's array, one at a time: the calling context owns it until it
 * has synchronised (ptx_synchronize, any read-back) or is destroyed, and a call from another context on that GPU meanwhile
 * fails with PTX_ERR_STATE instead of racing. */
int ptx_render_frames_noalloc(ptx_ctx *ctx, const int32_t *timestamps, uint32_t n_frames, uint32_t first_frame_number);
int ptx_synchronize(ptx_ctx *ctx);
/* Replaces gl.clear of the ping-pong targets (main.js:326-327) / starting a new progressive render. */
int ptx_reset_accum(ptx_ctx *ctx);
int ptx_frames_done(const ptx_ctx *ctx); /* frames accumulated so far (negative status on error) */

/* ---- read-back (replaces gl.readPixels, main.js:459) ---- */
/* rgb: W*H*3 floats, row 0 = bottom row like GL. Running mean of the accumulated frames. */
int ptx_read_linear_f32(ptx_ctx *ctx, float *rgb);
/* rgb: W*H*3 floats: the raw sum over frames (what the multi-GPU all-reduce adds up). */
int ptx_read_sum_f32(ptx_ctx *ctx, float *rgb);
/* out: W*H*4 bytes. apply_gamma: pow(c, 1/2.2) as shaders/fragmentShaderOutput.glsl:35 (pow = exp2(y * log2(x)) in the
 * operation sequence of the arithmetic contract); flip_y: the vertical flip of main.js:464-473
 * (row 0 = top row). Quantisation: floor(c*255 + 0.5). */
int ptx_read_rgba8(ptx_ctx *ctx, uint8_t *out, int apply_gamma, int flip_y);

/* The reference's RGBA8 ping-pong target after the last frame (option "compat_rgba8" = 1 or 2).  With apply_gamma it is
 * passed through the display pass (fragmentShaderOutput.glsl over the 8-bit target, main.js:357-373): what gl.readPixels
 * (main.js:459) returns in the browser, for frame-by-frame comparison with captures; the 8-bit store of the display pass
 * follows the same rule set as the option.  flip_y as ptx_read_rgba8. */
int ptx_read_compat_rgba8(ptx_ctx *ctx, uint8_t *out, int apply_gamma, int flip_y);
/* Checkpoint restore (new): sets the accumulation to rgb (W*H*3 floats as written by ptx_read_sum_f32) and
 * the frame count to frames_done, so a long progressive render (main.js:391-436 renderAsync) can resume after a restart. */
int ptx_write_sum_f32(ptx_ctx *ctx, const float *rgb, uint32_t frames_done);

/* ---- parity / roofline hooks (new) ---- */
/* Closest hit of n rays (o, d: 3n floats) by ray_bvh_intersection_hit_miss (glsl:207-249). A miss gives
 * t = 1e38f, tri_id = -1; a hit gives the ORIGINAL triangle index (inorder_tri[k]) of the winner. */
int ptx_trace_closest(ptx_ctx *ctx, const float *o, const float *d, uint32_t n, float *t, int32_t *tri_id);
/* Check of option "traversal" bit 0 (best-t pruning; the reference never prunes, glsl:207-249) under the rule option "prune"
 * selects -- a machine check beside the proof for prune = 1, the measure of what prune = 2 gives up: generates n_rays adversarial rays of one family ON THE DEVICE (0 surface -> any
 * direction, 1 surface -> surface, 2 in the own triangle's plane +- a tiny tilt, 3 grazing one of the 256 largest triangles
 * from afar, 4 skimming a triangle's bounding-box face from just outside its slab, 5 grazing hemisphere directions, 6 (nearly)
 * axis-parallel, 7 towards vertices and edges), traces each one with and without pruning through the tree the "traversal"
 * option selects (bit 1) with the arithmetic of the selected pipeline, and reports how many (t bits, triangle) results
 * differ, with the first 16 offenders (o.xyz, d.xyz, t exact, t pruned; triangle exact, triangle pruned).
 * tests/test_prune_certificate.py runs >= 1e8 rays per BASELINE scene: zero differences under prune = 1 and under the default
 * prune = 2; with margin_lat = 0 (round 2's margin rule) family 3 finds the grazing rays described at option "prune". */
typedef struct ptx_selfcheck_result {
    uint64_t rays, hits, mismatches;
    float first[16][8];
    int32_t first_tri[16][2]; /* -2 = unused entry */
    /* option "anyhit": rays whose unpruned closest hit is an emissive triangle, and how many of those the light bound
     * (the rule of option "prune" on the boxes around the emitters) would have let end on a blocker farther away.
     * Families 8 .. 11 are families 1, 3, 4, 7 aimed at the emissive triangles. */
    uint64_t emitter_hits, anyhit_violations;
} ptx_selfcheck_result;
int ptx_selfcheck_pruning(ptx_ctx *ctx, uint32_t family, uint64_t n_rays, uint32_t seed, ptx_selfcheck_result *out);
/* Primary-ray closest hits for every pixel (glsl:501-514 + 207-249); W*H entries, row 0 = bottom. */
int ptx_primary_hits(ptx_ctx *ctx, int32_t *tri_id, float *t);
/* Primary ray directions (glsl:145-148, 501-514); dirs: W*H*3 floats. */
int ptx_primary_dirs(ptx_ctx *ctx, float *dirs);
int ptx_get_counters(ptx_ctx *ctx, ptx_counters *out); /* synchronises */
int ptx_reset_counters(ptx_ctx *ctx);

/* ---- multi-GPU, one process per GPU (new; the reference has a single WebGL context, main.js:123) ----
 * Rank r of G renders frames {r, r + G, ...} with ptx_render_frames on its own single-device context; ptx_allreduce then
 * sums the accumulation buffers (W*H*4 f32; 33 MB at 1080p) and the frame counts of all ranks with one grouped
 * ncclAllReduce on the context's stream.  The communicator is RCCL's own: rank 0 calls ptx_comm_unique_id and hands the
 * 128 bytes to the other ranks over whatever channel the job has (torch.distributed, MPI, a file); every rank then calls
 * ptx_comm_init (collective: ncclCommInitRank).  ptx_comm_world returns the rank count RCCL reports (0 = no communicator;
 * the device count for a multi-device context). */
#define PTX_COMM_ID_BYTES 128
int ptx_comm_unique_id(void *id128);
int ptx_comm_init(ptx_ctx *ctx, const void *id128, int rank, int world);
int ptx_comm_world(const ptx_ctx *ctx);
/* d_dst == NULL: in place -- afterwards every rank's accumulation is the total and ptx_frames_done the total count, so the
 * read-backs give the same image on every rank (call it once, when the job's frames are rendered).  d_dst != NULL: the
 * total goes to that device buffer (W*H*4 floats) and the rank's own accumulation is left alone (progress previews).
 * frames_total (may be NULL) receives the summed frame count.  Blocks until the sum is there. */
int ptx_allreduce(ptx_ctx *ctx, void *d_dst, uint32_t *frames_total);

/* ---- reduction hooks for a caller that brings its own collective (new) ---- */
/* Device-to-device copy of the W*H*4 f32 accumulation buffer (rgb sum + unused 4th lane) to / from
 * caller-owned device memory (e.g. a torch tensor that torch.distributed all-reduces over RCCL). */
int ptx_accum_export_device(ptx_ctx *ctx, void *d_dst);
int ptx_accum_import_device(ptx_ctx *ctx, const void *d_src, uint32_t frames_done);

#ifdef __cplusplus
}
#endif
#endif /* PTX_H */
