/*
 * ptx_napi.c -- raw N-API (node_api.h) addon over the C ABI of include/ptx.h.
 *
 * This is the binding the JS host (path-tracing_amd/host/ptx.js, main.js) uses where the reference's main.js
 * talks to WebGL (uploadTexture / gl.uniform* / gl.drawArrays / gl.readPixels, main.js:272-355,459).
 * It only converts JS values <-> plain pointers and sizes: typed arrays are validated against T/N/L, every
 * non-zero ptx_status becomes a JS Error carrying ptx_last_error, results come back in addon-allocated
 * ArrayBuffers whose size comes from the context itself (ptx_get_size), never from the caller.  One context, one
 * thread: while a renderFramesAsync job runs on the libuv pool the handle is marked busy (every other call on it
 * throws, destroy included) and a reference keeps it alive until the job has settled.
 *
 * Build: cc -shared -fPIC -I/usr/include/node ptx_napi.c -o ptx_napi.node -L. -lptx_hip -Wl,-rpath,'$ORIGIN'
 */
#define NAPI_VERSION 6
#include <node_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "../../include/ptx.h"

#define NAPI_OK(env, call)                                                                  \
    do {                                                                                    \
        if ((call) != napi_ok) {                                                            \
            napi_throw_error((env), NULL, "ptx_napi: N-API call failed: " #call);           \
            return NULL;                                                                    \
        }                                                                                   \
    } while (0)

static napi_value throw_ptx(napi_env env, ptx_ctx *ctx, int rc) {
    char msg[600];
    snprintf(msg, sizeof msg, "ptx status %d: %s", rc, ptx_last_error(ctx));
    napi_throw_error(env, NULL, msg);
    return NULL;
}

static int get_args(napi_env env, napi_callback_info info, size_t want, napi_value *argv) {
    size_t argc = want;
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < want) {
        napi_throw_type_error(env, NULL, "ptx_napi: wrong number of arguments");
        return 0;
    }
    return 1;
}

typedef struct slot {
    ptx_ctx *ctx;
    int busy; /* a renderFramesAsync job owns the context until it settles */
} slot;

static slot *get_slot(napi_env env, napi_value v) {
    void *p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p || !((slot *)p)->ctx) {
        napi_throw_type_error(env, NULL, "ptx_napi: not a context handle (or already destroyed)");
        return NULL;
    }
    if (((slot *)p)->busy) {
        napi_throw_error(env, NULL, "ptx_napi: the context is busy with a renderFramesAsync job; await its promise first");
        return NULL;
    }
    return (slot *)p;
}

static ptx_ctx *get_ctx(napi_env env, napi_value v) {
    slot *s = get_slot(env, v);
    return s ? s->ctx : NULL;
}

/* size of every read-back buffer: what the context was last given, not what the caller believes */
static int get_size(napi_env env, ptx_ctx *ctx, uint32_t *w, uint32_t *h) {
    if (ptx_get_size(ctx, w, h) != PTX_OK) {
        napi_throw_error(env, NULL, "ptx_napi: setParams first");
        return 0;
    }
    return 1;
}

/* typed array of the given type with exactly `want` elements (want = SIZE_MAX: any length) */
static void *get_typed(napi_env env, napi_value v, napi_typedarray_type type, size_t want, size_t *len_out, const char *name) {
    bool is = false;
    napi_typedarray_type t;
    size_t len = 0;
    void *data = NULL;
    char msg[200];
    if (napi_is_typedarray(env, v, &is) != napi_ok || !is || napi_get_typedarray_info(env, v, &t, &len, &data, NULL, NULL) != napi_ok || t != type) {
        snprintf(msg, sizeof msg, "ptx_napi: %s must be a %s", name, type == napi_float32_array ? "Float32Array" : "Int32Array");
        napi_throw_type_error(env, NULL, msg);
        return NULL;
    }
    if (want != SIZE_MAX && len != want) {
        snprintf(msg, sizeof msg, "ptx_napi: %s has %zu elements, expected %zu", name, len, want);
        napi_throw_range_error(env, NULL, msg);
        return NULL;
    }
    if (len_out) *len_out = len;
    static float dummy;
    return data ? data : (void *)&dummy; /* zero-length arrays may report NULL */
}

static int get_prop_u32(napi_env env, napi_value obj, const char *name, uint32_t *out) {
    napi_value v;
    if (napi_get_named_property(env, obj, name, &v) != napi_ok || napi_get_value_uint32(env, v, out) != napi_ok) {
        char msg[120];
        snprintf(msg, sizeof msg, "ptx_napi: property %s must be a non-negative integer", name);
        napi_throw_type_error(env, NULL, msg);
        return 0;
    }
    return 1;
}

static int get_prop_i32(napi_env env, napi_value obj, const char *name, int32_t *out) {
    napi_value v;
    if (napi_get_named_property(env, obj, name, &v) != napi_ok || napi_get_value_int32(env, v, out) != napi_ok) {
        char msg[120];
        snprintf(msg, sizeof msg, "ptx_napi: property %s must be an integer", name);
        napi_throw_type_error(env, NULL, msg);
        return 0;
    }
    return 1;
}

/* JS numbers are f64; they are rounded to f32 here, once, like gl.uniform3f does (main.js:292-296) */
static int get_prop_vec3(napi_env env, napi_value obj, const char *name, float out[3]) {
    napi_value v, e;
    if (napi_get_named_property(env, obj, name, &v) != napi_ok) goto bad;
    for (uint32_t i = 0; i < 3; i++) {
        double d;
        if (napi_get_element(env, v, i, &e) != napi_ok || napi_get_value_double(env, e, &d) != napi_ok) goto bad;
        out[i] = (float)d;
    }
    return 1;
bad: {
    char msg[120];
    snprintf(msg, sizeof msg, "ptx_napi: property %s must be an array of 3 numbers", name);
    napi_throw_type_error(env, NULL, msg);
    return 0;
}
}

static napi_value make_typed(napi_env env, napi_typedarray_type type, size_t n, size_t elem, void **data) {
    napi_value ab, ta;
    if (napi_create_arraybuffer(env, n * elem, data, &ab) != napi_ok || napi_create_typedarray(env, type, n, ab, 0, &ta) != napi_ok) {
        napi_throw_error(env, NULL, "ptx_napi: allocation failed");
        return NULL;
    }
    return ta;
}

static void finalize_ctx(napi_env env, void *data, void *hint) {
    (void)env; (void)hint;
    slot *s = (slot *)data;
    if (s->ctx) ptx_destroy(s->ctx);
    free(s);
}

static napi_value DeviceCount(napi_env env, napi_callback_info info) {
    (void)info;
    napi_value r;
    NAPI_OK(env, napi_create_int32(env, ptx_device_count(), &r));
    return r;
}

static napi_value Create(napi_env env, napi_callback_info info) {
    /* create(device) or create([device, ...]): one context over one or several GPUs (ptx_create) */
    napi_value argv[1], ext, e;
    int32_t devs[64];
    uint32_t n = 1;
    bool is_array = false;
    if (!get_args(env, info, 1, argv)) return NULL;
    if (napi_is_array(env, argv[0], &is_array) == napi_ok && is_array) {
        if (napi_get_array_length(env, argv[0], &n) != napi_ok || n < 1 || n > 64) { napi_throw_range_error(env, NULL, "ptx_napi: 1 .. 64 devices"); return NULL; }
        for (uint32_t i = 0; i < n; i++)
            if (napi_get_element(env, argv[0], i, &e) != napi_ok || napi_get_value_int32(env, e, &devs[i]) != napi_ok) { napi_throw_type_error(env, NULL, "ptx_napi: devices must be integers"); return NULL; }
    } else if (napi_get_value_int32(env, argv[0], &devs[0]) != napi_ok) { napi_throw_type_error(env, NULL, "ptx_napi: device must be an integer or an array of integers"); return NULL; }
    ptx_ctx *ctx = ptx_create(devs, (int)n);
    if (!ctx) return throw_ptx(env, NULL, PTX_ERR_HIP);
    slot *s = (slot *)calloc(1, sizeof *s);
    if (!s) { ptx_destroy(ctx); napi_throw_error(env, NULL, "ptx_napi: out of memory"); return NULL; }
    s->ctx = ctx;
    if (napi_create_external(env, s, finalize_ctx, NULL, &ext) != napi_ok) { ptx_destroy(ctx); free(s); napi_throw_error(env, NULL, "ptx_napi: cannot create the handle"); return NULL; }
    return ext;
}

static napi_value Destroy(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    void *p = NULL;
    if (!get_args(env, info, 1, argv)) return NULL;
    if (napi_get_value_external(env, argv[0], &p) == napi_ok && p && ((slot *)p)->ctx) {
        if (((slot *)p)->busy) { napi_throw_error(env, NULL, "ptx_napi: destroy() while a renderFramesAsync job runs; await its promise first"); return NULL; }
        ptx_destroy(((slot *)p)->ctx);
        ((slot *)p)->ctx = NULL;
    }
    return NULL;
}

static napi_value UploadScene(napi_env env, napi_callback_info info) {
    napi_value argv[2], v;
    if (!get_args(env, info, 2, argv)) return NULL;
    ptx_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    ptx_scene s;
    memset(&s, 0, sizeof s);
    if (!get_prop_u32(env, argv[1], "T", &s.T) || !get_prop_u32(env, argv[1], "N", &s.N) || !get_prop_u32(env, argv[1], "L", &s.L)) return NULL;
#define FARR(field, prop, n)                                                                           \
    if (napi_get_named_property(env, argv[1], prop, &v) != napi_ok) return NULL;                       \
    if (!(s.field = (const float *)get_typed(env, v, napi_float32_array, (size_t)(n), NULL, prop))) return NULL;
#define IARR(field, prop, n)                                                                           \
    if (napi_get_named_property(env, argv[1], prop, &v) != napi_ok) return NULL;                       \
    if (!(s.field = (const int32_t *)get_typed(env, v, napi_int32_array, (size_t)(n), NULL, prop))) return NULL;
    FARR(coords, "coords", 9 * (size_t)s.T) FARR(normals, "normals", 3 * (size_t)s.T) FARR(colors, "colors", 3 * (size_t)s.T)
    FARR(emissions, "emissions", 3 * (size_t)s.T) IARR(light_tri, "lightTri", s.L) FARR(node_min, "nodeMin", 3 * (size_t)s.N)
    FARR(node_max, "nodeMax", 3 * (size_t)s.N) IARR(node_miss, "nodeMiss", s.N) IARR(node_first, "nodeFirst", s.N)
    IARR(node_count, "nodeCount", s.N) IARR(inorder_tri, "inorderTri", s.T)
#undef FARR
#undef IARR
    int rc = ptx_upload_scene(ctx, &s);
    if (rc) return throw_ptx(env, ctx, rc);
    return NULL;
}

static napi_value SetParams(napi_env env, napi_callback_info info) {
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return NULL;
    ptx_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    ptx_params p;
    memset(&p, 0, sizeof p);
    if (!get_prop_u32(env, argv[1], "width", &p.width) || !get_prop_u32(env, argv[1], "height", &p.height) ||
        !get_prop_vec3(env, argv[1], "camSrc", p.cam_src) || !get_prop_vec3(env, argv[1], "camUp", p.cam_up) ||
        !get_prop_vec3(env, argv[1], "camRight", p.cam_right) || !get_prop_vec3(env, argv[1], "camLeftBottom", p.cam_left_bottom) ||
        !get_prop_i32(env, argv[1], "maxPathLength", &p.max_path_length) || !get_prop_i32(env, argv[1], "sampleCount", &p.sample_count))
        return NULL;
    int rc = ptx_set_params(ctx, &p);
    if (rc) return throw_ptx(env, ctx, rc);
    return NULL;
}

static napi_value SetOption(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    char key[64];
    size_t n = 0;
    int64_t val = 0;
    if (!get_args(env, info, 3, argv)) return NULL;
    ptx_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    if (napi_get_value_string_utf8(env, argv[1], key, sizeof key, &n) != napi_ok || napi_get_value_int64(env, argv[2], &val) != napi_ok) {
        napi_throw_type_error(env, NULL, "ptx_napi: setOption(handle, string, integer)");
        return NULL;
    }
    int rc = ptx_set_option(ctx, key, val);
    if (rc) return throw_ptx(env, ctx, rc);
    return NULL;
}

static napi_value RenderFrames(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    size_t n = 0;
    uint32_t first = 0;
    if (!get_args(env, info, 3, argv)) return NULL;
    ptx_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    const int32_t *ts = (const int32_t *)get_typed(env, argv[1], napi_int32_array, SIZE_MAX, &n, "timestamps");
    if (!ts) return NULL;
    if (napi_get_value_uint32(env, argv[2], &first) != napi_ok) { napi_throw_type_error(env, NULL, "ptx_napi: firstFrameNumber must be an integer"); return NULL; }
    int rc = ptx_render_frames(ctx, ts, (uint32_t)n, first);
    if (rc == PTX_OK) rc = ptx_synchronize(ctx); /* the typed array may be collected once we return */
    if (rc) return throw_ptx(env, ctx, rc);
    return NULL;
}

/* renderFramesAsync(handle, Int32Array timestamps, firstFrameNumber) -> Promise<undefined>.
 * The frame loop of the reference is async (main.js:262,391: the page stays responsive while frames render); this is the
 * counterpart: the frames are queued and awaited on a libuv worker thread, the promise settles on the JS thread.  The
 * context is not thread-safe: no other call on the same handle until the promise has settled. */
typedef struct render_job {
    napi_async_work work;
    napi_deferred deferred;
    napi_ref handle_ref; /* keeps the external handle (and with it the context) alive while the worker runs */
    slot *owner;
    ptx_ctx *ctx;
    int32_t *ts;
    uint32_t n, first;
    int rc;
    char msg[600];
} render_job;

static void render_execute(napi_env env, void *data) {
    render_job *j = (render_job *)data;
    (void)env;
    j->rc = ptx_render_frames(j->ctx, j->ts, j->n, j->first);
    if (j->rc == PTX_OK) j->rc = ptx_synchronize(j->ctx);
    if (j->rc) snprintf(j->msg, sizeof j->msg, "ptx status %d: %s", j->rc, ptx_last_error(j->ctx));
}

static void render_complete(napi_env env, napi_status status, void *data) {
    render_job *j = (render_job *)data;
    napi_value v;
    if (status != napi_ok || j->rc) {
        napi_value m;
        napi_create_string_utf8(env, j->rc ? j->msg : "ptx_napi: render job was cancelled", NAPI_AUTO_LENGTH, &m);
        napi_create_error(env, NULL, m, &v);
        napi_reject_deferred(env, j->deferred, v);
    } else {
        napi_get_undefined(env, &v);
        napi_resolve_deferred(env, j->deferred, v);
    }
    j->owner->busy = 0;
    napi_delete_reference(env, j->handle_ref);
    napi_delete_async_work(env, j->work);
    free(j->ts);
    free(j);
}

static napi_value RenderFramesAsync(napi_env env, napi_callback_info info) {
    napi_value argv[3], promise, name;
    size_t n = 0;
    uint32_t first = 0;
    if (!get_args(env, info, 3, argv)) return NULL;
    slot *owner = get_slot(env, argv[0]);
    if (!owner) return NULL;
    ptx_ctx *ctx = owner->ctx;
    const int32_t *ts = (const int32_t *)get_typed(env, argv[1], napi_int32_array, SIZE_MAX, &n, "timestamps");
    if (!ts) return NULL;
    if (napi_get_value_uint32(env, argv[2], &first) != napi_ok) { napi_throw_type_error(env, NULL, "ptx_napi: firstFrameNumber must be an integer"); return NULL; }
    render_job *j = (render_job *)calloc(1, sizeof *j);
    if (j) j->ts = (int32_t *)malloc((n ? n : 1) * sizeof(int32_t));
    if (!j || !j->ts) { free(j); napi_throw_error(env, NULL, "ptx_napi: out of memory"); return NULL; }
    memcpy(j->ts, ts, n * sizeof(int32_t));     /* the typed array may be collected before the worker runs */
    j->ctx = ctx; j->owner = owner; j->n = (uint32_t)n; j->first = first;
    if (napi_create_reference(env, argv[0], 1, &j->handle_ref) != napi_ok) { free(j->ts); free(j); napi_throw_error(env, NULL, "ptx_napi: cannot reference the handle"); return NULL; }
    owner->busy = 1;
    if (napi_create_promise(env, &j->deferred, &promise) != napi_ok ||
        napi_create_string_utf8(env, "ptx_render_frames", NAPI_AUTO_LENGTH, &name) != napi_ok ||
        napi_create_async_work(env, NULL, name, render_execute, render_complete, j, &j->work) != napi_ok ||
        napi_queue_async_work(env, j->work) != napi_ok) {
        owner->busy = 0;
        napi_delete_reference(env, j->handle_ref);
        free(j->ts); free(j);
        napi_throw_error(env, NULL, "ptx_napi: cannot queue the render job");
        return NULL;
    }
    return promise;
}

static napi_value ResetAccum(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    ptx_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    int rc = ptx_reset_accum(ctx);
    if (rc) return throw_ptx(env, ctx, rc);
    return NULL;
}

static napi_value FramesDone(napi_env env, napi_callback_info info) {
    napi_value argv[1], r;
    if (!get_args(env, info, 1, argv)) return NULL;
    ptx_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    NAPI_OK(env, napi_create_int32(env, ptx_frames_done(ctx), &r));
    return r;
}

static napi_value read_f32(napi_env env, napi_callback_info info, int (*fn)(ptx_ctx *, float *)) {
    napi_value argv[1];
    uint32_t w = 0, h = 0;
    void *data = NULL;
    if (!get_args(env, info, 1, argv)) return NULL;
    ptx_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx || !get_size(env, ctx, &w, &h)) return NULL;
    napi_value ta = make_typed(env, napi_float32_array, (size_t)w * h * 3, 4, &data);
    if (!ta) return NULL;
    int rc = fn(ctx, (float *)data);
    if (rc) return throw_ptx(env, ctx, rc);
    return ta;
}
static napi_value ReadLinear(napi_env env, napi_callback_info info) { return read_f32(env, info, ptx_read_linear_f32); }
static napi_value ReadSum(napi_env env, napi_callback_info info) { return read_f32(env, info, ptx_read_sum_f32); }

static napi_value read_u8(napi_env env, napi_callback_info info, int (*fn)(ptx_ctx *, uint8_t *, int, int), const char *usage) {
    napi_value argv[3];
    uint32_t w = 0, h = 0;
    bool gamma = true, flip = true;
    void *data = NULL;
    if (!get_args(env, info, 3, argv)) return NULL;
    ptx_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx || !get_size(env, ctx, &w, &h)) return NULL;
    if (napi_get_value_bool(env, argv[1], &gamma) != napi_ok || napi_get_value_bool(env, argv[2], &flip) != napi_ok) {
        napi_throw_type_error(env, NULL, usage);
        return NULL;
    }
    napi_value ta = make_typed(env, napi_uint8_array, (size_t)w * h * 4, 1, &data);
    if (!ta) return NULL;
    int rc = fn(ctx, (uint8_t *)data, gamma, flip);
    if (rc) return throw_ptx(env, ctx, rc);
    return ta;
}
static napi_value ReadRgba8(napi_env env, napi_callback_info info) {
    return read_u8(env, info, ptx_read_rgba8, "ptx_napi: readRgba8(handle, gamma:boolean, flipY:boolean)");
}
static napi_value ReadCompatRgba8(napi_env env, napi_callback_info info) {
    return read_u8(env, info, ptx_read_compat_rgba8, "ptx_napi: readCompatRgba8(handle, gamma:boolean, flipY:boolean)");
}

/* writeSum(handle, Float32Array rgb (W*H*3 of the context's size), framesDone): checkpoint restore */
static napi_value WriteSum(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    uint32_t w = 0, h = 0, frames = 0;
    if (!get_args(env, info, 3, argv)) return NULL;
    ptx_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx || !get_size(env, ctx, &w, &h)) return NULL;
    if (napi_get_value_uint32(env, argv[2], &frames) != napi_ok) {
        napi_throw_type_error(env, NULL, "ptx_napi: writeSum(handle, rgb:Float32Array, framesDone)");
        return NULL;
    }
    const float *rgb = (const float *)get_typed(env, argv[1], napi_float32_array, (size_t)w * h * 3, NULL, "rgb");
    if (!rgb) return NULL;
    int rc = ptx_write_sum_f32(ctx, rgb, frames);
    if (rc) return throw_ptx(env, ctx, rc);
    return NULL;
}

static napi_value pair_result(napi_env env, napi_value t, napi_value tri) {
    napi_value o;
    NAPI_OK(env, napi_create_object(env, &o));
    NAPI_OK(env, napi_set_named_property(env, o, "t", t));
    NAPI_OK(env, napi_set_named_property(env, o, "tri", tri));
    return o;
}

static napi_value TraceClosest(napi_env env, napi_callback_info info) {
    napi_value argv[3];
    size_t no = 0, nd = 0;
    void *t = NULL, *tri = NULL;
    if (!get_args(env, info, 3, argv)) return NULL;
    ptx_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    const float *o = (const float *)get_typed(env, argv[1], napi_float32_array, SIZE_MAX, &no, "origins");
    if (!o) return NULL;
    const float *d = (const float *)get_typed(env, argv[2], napi_float32_array, SIZE_MAX, &nd, "directions");
    if (!d) return NULL;
    if (no != nd || no % 3) { napi_throw_range_error(env, NULL, "ptx_napi: origins and directions must both hold 3n floats"); return NULL; }
    napi_value ta_t = make_typed(env, napi_float32_array, no / 3, 4, &t), ta_tri = make_typed(env, napi_int32_array, no / 3, 4, &tri);
    if (!ta_t || !ta_tri) return NULL;
    int rc = ptx_trace_closest(ctx, o, d, (uint32_t)(no / 3), (float *)t, (int32_t *)tri);
    if (rc) return throw_ptx(env, ctx, rc);
    return pair_result(env, ta_t, ta_tri);
}

static napi_value PrimaryHits(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    uint32_t w = 0, h = 0;
    void *t = NULL, *tri = NULL;
    if (!get_args(env, info, 1, argv)) return NULL;
    ptx_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx || !get_size(env, ctx, &w, &h)) return NULL;
    napi_value ta_t = make_typed(env, napi_float32_array, (size_t)w * h, 4, &t), ta_tri = make_typed(env, napi_int32_array, (size_t)w * h, 4, &tri);
    if (!ta_t || !ta_tri) return NULL;
    int rc = ptx_primary_hits(ctx, (int32_t *)tri, (float *)t);
    if (rc) return throw_ptx(env, ctx, rc);
    return pair_result(env, ta_t, ta_tri);
}

static napi_value Counters(napi_env env, napi_callback_info info) {
    napi_value argv[1], o, v;
    ptx_counters c;
    if (!get_args(env, info, 1, argv)) return NULL;
    ptx_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    int rc = ptx_get_counters(ctx, &c);
    if (rc) return throw_ptx(env, ctx, rc);
    NAPI_OK(env, napi_create_object(env, &o));
#define PUT(name, val) NAPI_OK(env, napi_create_double(env, (double)(val), &v)); NAPI_OK(env, napi_set_named_property(env, o, name, v));
    PUT("samples", c.samples) PUT("rays", c.rays) PUT("raysTraced", c.rays_traced) PUT("nodeVisits", c.node_visits) PUT("triTests", c.tri_tests)
    PUT("rayHits", c.ray_hits) PUT("lightAttempts", c.light_attempts) PUT("pixelFrames", c.pixel_frames) PUT("kernelLaunches", c.kernel_launches)
    PUT("kernelMs", c.kernel_ms)
#undef PUT
    return o;
}

static napi_value CommWorld(napi_env env, napi_callback_info info) {
    napi_value argv[1], r;
    if (!get_args(env, info, 1, argv)) return NULL;
    ptx_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    NAPI_OK(env, napi_create_int32(env, ptx_comm_world(ctx), &r));
    return r;
}

static napi_value ResetCounters(napi_env env, napi_callback_info info) {
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return NULL;
    ptx_ctx *ctx = get_ctx(env, argv[0]);
    if (!ctx) return NULL;
    int rc = ptx_reset_counters(ctx);
    if (rc) return throw_ptx(env, ctx, rc);
    return NULL;
}

static napi_value Init(napi_env env, napi_value exports) {
    static const struct { const char *name; napi_callback fn; } fns[] = {
        {"deviceCount", DeviceCount}, {"create", Create}, {"destroy", Destroy}, {"uploadScene", UploadScene}, {"setParams", SetParams},
        {"setOption", SetOption}, {"renderFrames", RenderFrames}, {"renderFramesAsync", RenderFramesAsync}, {"resetAccum", ResetAccum}, {"framesDone", FramesDone},
        {"readLinear", ReadLinear}, {"readSum", ReadSum}, {"readRgba8", ReadRgba8}, {"readCompatRgba8", ReadCompatRgba8}, {"writeSum", WriteSum}, {"traceClosest", TraceClosest},
        {"primaryHits", PrimaryHits}, {"counters", Counters}, {"commWorld", CommWorld}, {"resetCounters", ResetCounters}};
    for (size_t i = 0; i < sizeof fns / sizeof fns[0]; i++) {
        napi_value f;
        if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok ||
            napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) {
            napi_throw_error(env, NULL, "ptx_napi: init failed");
            return NULL;
        }
    }
    return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
