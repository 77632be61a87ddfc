/*
 * gs_napi.c -- plain-C N-API shim over include/gsplat/gs_abi.h (NAPI v4+, Node >= 12).
 *
 * This is the binding a Node host uses to put the MI355X rasterizer behind the reference's
 * TypeScript surface (Renderer / Camera / PackedGaussians, see ../../js).  It adds nothing to the
 * C ABI: every export is a 1:1 wrapper; HIP errors become thrown JS Errors / rejected Promises.
 * The per-frame call never blocks the JS thread: renderAsync() enqueues and waits in a libuv worker
 * (napi_async_work) and resolves a Promise, which is what Renderer.animate() awaits where the
 * reference awaits queue.onSubmittedWorkDone() (renderer.ts:404-587).
 */
#define NAPI_VERSION 4
#include <node_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../../include/gsplat/gs_abi.h"

#define NAPI_CALL(env, call)                                              \
    do {                                                                  \
        napi_status s_ = (call);                                          \
        if (s_ != napi_ok) {                                              \
            napi_throw_error((env), NULL, "N-API call failed: " #call);   \
            return NULL;                                                  \
        }                                                                 \
    } while (0)

static napi_value throw_gs(napi_env env, int32_t rc) {
    char msg[640];
    const char* e = gs_last_error();
    strcpy(msg, "gsplat: ");
    strncat(msg, e && e[0] ? e : "error", sizeof(msg) - 32);
    char code[16];
    int n = 0, v = rc < 0 ? -rc : rc;
    code[n++] = '-';
    if (v >= 10) code[n++] = (char)('0' + v / 10);
    code[n++] = (char)('0' + v % 10);
    code[n] = 0;
    napi_throw_error(env, code, msg);
    return NULL;
}

static int get_u32_prop(napi_env env, napi_value obj, const char* name, uint32_t* out) {
    napi_value v;
    bool has = false;
    if (napi_has_named_property(env, obj, name, &has) != napi_ok || !has) return 0;
    if (napi_get_named_property(env, obj, name, &v) != napi_ok) return 0;
    double d;
    if (napi_get_value_double(env, v, &d) != napi_ok) return 0;
    *out = (uint32_t)d;
    return 1;
}

/* The external holds a box so that destroy() can null the context.  A gs_ctx is not re-entrant and renderAsync() works on it
 * from a libuv worker: while a frame is in flight (`busy`) every other call on the handle is refused, and destroy() is
 * deferred to the frame's completion (all of this happens on the JS thread, so the flags need no lock). */
typedef struct {
    gs_ctx* ctx;
    int busy;            /* a renderAsync job owns the context */
    int inflight;        /* renderToSink frames whose tickets are being waited for on workers (they only call gs_wait_ticket) */
    int destroy_pending; /* destroy() was called meanwhile */
} ctx_box;

static ctx_box* unbox(napi_env env, napi_value v) {
    void* p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
        napi_throw_type_error(env, NULL, "gsplat: expected a context handle");
        return NULL;
    }
    return (ctx_box*)p;
}

static gs_ctx* unwrap(napi_env env, napi_value v) {
    ctx_box* b = unbox(env, v);
    if (!b) return NULL;
    if (!b->ctx) {
        napi_throw_error(env, NULL, "gsplat: the context has been destroyed");
        return NULL;
    }
    if (b->busy || b->inflight) {
        napi_throw_error(env, NULL, "gsplat: a frame is in flight on this context (await renderAsync / the renderToSink promises first)");
        return NULL;
    }
    return b->ctx;
}

static void finalize_ctx(napi_env env, void* data, void* hint) {
    (void)env; (void)hint;
    ctx_box* box = (ctx_box*)data;
    if (box->busy || box->inflight) { box->destroy_pending = 2; return; } /* the last job frees the box when it completes */
    if (box->ctx) gs_destroy(box->ctx);
    free(box);
}

/* bytes of an ArrayBuffer or of any TypedArray/Buffer view */
static int get_bytes(napi_env env, napi_value v, void** data, size_t* len) {
    bool is = false;
    if (napi_is_arraybuffer(env, v, &is) == napi_ok && is) return napi_get_arraybuffer_info(env, v, data, len) == napi_ok;
    if (napi_is_typedarray(env, v, &is) == napi_ok && is) {
        napi_typedarray_type t;
        size_t n, off;
        napi_value ab;
        if (napi_get_typedarray_info(env, v, &t, &n, data, &ab, &off) != napi_ok) return 0;
        static const size_t sz[] = {1, 1, 1, 2, 2, 4, 4, 4, 8, 8, 8};
        *len = n * sz[t];
        return 1;
    }
    if (napi_is_buffer(env, v, &is) == napi_ok && is) return napi_get_buffer_info(env, v, data, len) == napi_ok;
    return 0;
}

/* create({width,height,tileSize,device,colBegin,colEnd,flags,maxIntersections}) -> handle */
static napi_value js_create(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    gs_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = sizeof(cfg);
    cfg.tile_size = 16;
    uint32_t dev = 0, maxi = 0;
    if (argc < 1 || !get_u32_prop(env, argv[0], "width", &cfg.width) || !get_u32_prop(env, argv[0], "height", &cfg.height)) {
        napi_throw_type_error(env, NULL, "gsplat.create: {width, height} required");
        return NULL;
    }
    get_u32_prop(env, argv[0], "tileSize", &cfg.tile_size);
    if (get_u32_prop(env, argv[0], "device", &dev)) cfg.device = (int32_t)dev;
    get_u32_prop(env, argv[0], "colBegin", &cfg.col_begin);
    get_u32_prop(env, argv[0], "colEnd", &cfg.col_end);
    get_u32_prop(env, argv[0], "flags", &cfg.flags);
    if (get_u32_prop(env, argv[0], "maxIntersections", &maxi)) cfg.max_intersections = maxi;
    gs_ctx* ctx = NULL;
    int32_t rc = gs_create(&cfg, &ctx);
    if (rc != GS_OK) return throw_gs(env, rc);
    ctx_box* box = (ctx_box*)calloc(1, sizeof(ctx_box));
    if (!box) { gs_destroy(ctx); napi_throw_error(env, NULL, "gsplat.create: out of memory"); return NULL; }
    box->ctx = ctx;
    napi_value ext;
    if (napi_create_external(env, box, finalize_ctx, NULL, &ext) != napi_ok) {
        gs_destroy(ctx);
        free(box);
        napi_throw_error(env, NULL, "N-API call failed: napi_create_external");
        return NULL;
    }
    return ext;
}

static napi_value js_destroy(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    void* p = NULL;
    if (argc < 1 || napi_get_value_external(env, argv[0], &p) != napi_ok || !p) return NULL;
    ctx_box* box = (ctx_box*)p;
    if (box->busy || box->inflight) { /* a frame is in flight on a worker: destroy when it completes */
        if (!box->destroy_pending) box->destroy_pending = 1;
        return NULL;
    }
    if (box->ctx) {
        gs_destroy(box->ctx);
        box->ctx = NULL;
    }
    return NULL;
}

/* uploadSplats(handle, bytes, n) */
static napi_value js_upload(napi_env env, napi_callback_info info) {
    size_t argc = 3;
    napi_value argv[3];
    NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    gs_ctx* ctx = argc >= 3 ? unwrap(env, argv[0]) : NULL;
    if (!ctx) return NULL;
    void* data = NULL;
    size_t len = 0;
    double n = 0;
    if (!get_bytes(env, argv[1], &data, &len) || napi_get_value_double(env, argv[2], &n) != napi_ok || !(n >= 0.0) ||
        n != (double)(uint64_t)n || n >= 2147483648.0 || (double)len < n * GS_SPLAT_RECORD_BYTES) {
        napi_throw_type_error(env, NULL, "gsplat.uploadSplats: need n*320 bytes");
        return NULL;
    }
    int32_t rc = gs_upload_splats(ctx, data, (uint64_t)n);
    return rc == GS_OK ? NULL : throw_gs(env, rc);
}

/* shareSplats(handle, ownerHandle): render the owner's resident splats from a second context (gs_share_splats) */
static napi_value js_share(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    gs_ctx* ctx = argc >= 2 ? unwrap(env, argv[0]) : NULL;
    if (!ctx) return NULL;
    gs_ctx* owner = unwrap(env, argv[1]);
    if (!owner) return NULL;
    int32_t rc = gs_share_splats(ctx, owner);
    return rc == GS_OK ? NULL : throw_gs(env, rc);
}

/* renderSync(handle, uniforms160[, debug]) : enqueue + wait on the calling thread */
static napi_value js_render_sync(napi_env env, napi_callback_info info) {
    size_t argc = 3;
    napi_value argv[3];
    NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    gs_ctx* ctx = argc >= 2 ? unwrap(env, argv[0]) : NULL;
    if (!ctx) return NULL;
    void* data = NULL;
    size_t len = 0;
    if (!get_bytes(env, argv[1], &data, &len) || len < GS_UNIFORM_BYTES) {
        napi_throw_type_error(env, NULL, "gsplat.render: need the 160-byte uniform block");
        return NULL;
    }
    bool debug = false;
    if (argc >= 3) napi_get_value_bool(env, argv[2], &debug);
    int32_t rc = debug ? gs_render_debug(ctx, data) : gs_render(ctx, data);
    if (rc == GS_OK) rc = gs_wait(ctx);
    return rc == GS_OK ? NULL : throw_gs(env, rc);
}

typedef struct {
    napi_async_work work;
    napi_deferred deferred;
    ctx_box* box;
    gs_ctx* ctx;
    unsigned char uniforms[GS_UNIFORM_BYTES];
    int32_t rc;
    char err[512];
} frame_job;

static void job_execute(napi_env env, void* data) {
    (void)env;
    frame_job* j = (frame_job*)data;
    j->rc = gs_render(j->ctx, j->uniforms);
    if (j->rc == GS_OK) j->rc = gs_wait(j->ctx);
    if (j->rc != GS_OK) { /* gs_last_error is thread-local: capture it on the worker thread */
        strncpy(j->err, gs_last_error(), sizeof(j->err) - 1);
        j->err[sizeof(j->err) - 1] = 0;
    }
}

static void job_complete(napi_env env, napi_status status, void* data) {
    frame_job* j = (frame_job*)data;
    napi_value v;
    ctx_box* box = j->box;
    box->busy = 0;
    if (box->destroy_pending) { /* destroy() (1) or the finalizer (2) came while the frame was in flight */
        if (box->ctx) gs_destroy(box->ctx);
        box->ctx = NULL;
        if (box->destroy_pending == 2) free(box);
        else box->destroy_pending = 0;
    }
    if (status == napi_ok && j->rc == GS_OK) {
        napi_get_undefined(env, &v);
        napi_resolve_deferred(env, j->deferred, v);
    } else {
        napi_value msg;
        napi_create_string_utf8(env, j->rc != GS_OK ? j->err : "gsplat: async work cancelled", NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, NULL, msg, &v);
        napi_reject_deferred(env, j->deferred, v);
    }
    napi_delete_async_work(env, j->work);
    free(j);
}

/* renderAsync(handle, uniforms160) -> Promise<void> */
static napi_value js_render_async(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    gs_ctx* ctx = argc >= 2 ? unwrap(env, argv[0]) : NULL;
    if (!ctx) return NULL;
    void* data = NULL;
    size_t len = 0;
    if (!get_bytes(env, argv[1], &data, &len) || len < GS_UNIFORM_BYTES) {
        napi_throw_type_error(env, NULL, "gsplat.renderAsync: need the 160-byte uniform block");
        return NULL;
    }
    frame_job* j = (frame_job*)calloc(1, sizeof(frame_job));
    if (!j) { napi_throw_error(env, NULL, "gsplat.renderAsync: out of memory"); return NULL; }
    j->box = unbox(env, argv[0]);
    j->ctx = ctx;
    memcpy(j->uniforms, data, GS_UNIFORM_BYTES);
    napi_value promise, name;
    if (napi_create_promise(env, &j->deferred, &promise) != napi_ok ||
        napi_create_string_utf8(env, "gsplat.frame", NAPI_AUTO_LENGTH, &name) != napi_ok ||
        napi_create_async_work(env, NULL, name, job_execute, job_complete, j, &j->work) != napi_ok) {
        free(j);
        napi_throw_error(env, NULL, "gsplat.renderAsync: N-API call failed");
        return NULL;
    }
    j->box->busy = 1;
    if (napi_queue_async_work(env, j->work) != napi_ok) {
        j->box->busy = 0;
        napi_delete_async_work(env, j->work);
        free(j);
        napi_throw_error(env, NULL, "gsplat.renderAsync: napi_queue_async_work failed");
        return NULL;
    }
    return promise;
}

/* readRgba8(handle) -> ArrayBuffer (height*slabWidth*4) */
static napi_value js_read_rgba8(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    gs_ctx* ctx = argc >= 1 ? unwrap(env, argv[0]) : NULL;
    if (!ctx) return NULL;
    uint64_t bytes = 0;
    int32_t rc = gs_read_buffer(ctx, GS_BUF_RGBA8, NULL, 0, &bytes);
    if (rc != GS_OK) return throw_gs(env, rc);
    void* dst = NULL;
    napi_value ab;
    NAPI_CALL(env, napi_create_arraybuffer(env, (size_t)bytes, &dst, &ab));
    rc = gs_read_rgba8(ctx, dst, bytes);
    return rc == GS_OK ? ab : throw_gs(env, rc);
}

/* readBuffer(handle, which) -> ArrayBuffer */
static napi_value js_read_buffer(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    gs_ctx* ctx = argc >= 2 ? unwrap(env, argv[0]) : NULL;
    if (!ctx) return NULL;
    int32_t which = 0;
    NAPI_CALL(env, napi_get_value_int32(env, argv[1], &which));
    uint64_t bytes = 0;
    int32_t rc = gs_read_buffer(ctx, which, NULL, 0, &bytes);
    if (rc != GS_OK) return throw_gs(env, rc);
    void* dst = NULL;
    napi_value ab;
    NAPI_CALL(env, napi_create_arraybuffer(env, (size_t)bytes, &dst, &ab));
    if (bytes) rc = gs_read_buffer(ctx, which, dst, bytes, NULL);
    return rc == GS_OK ? ab : throw_gs(env, rc);
}

static void set_num(napi_env env, napi_value obj, const char* k, double v) {
    napi_value n;
    napi_create_double(env, v, &n);
    napi_set_named_property(env, obj, k, n);
}

/* stats(handle) -> {numGaussians, numVisible, numIntersections, numProcessed, numTiles, sortPasses, frames, stageUs:[6], frameUs} */
static napi_value js_stats(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    gs_ctx* ctx = argc >= 1 ? unwrap(env, argv[0]) : NULL;
    if (!ctx) return NULL;
    gs_stats st;
    int32_t rc = gs_get_stats(ctx, &st);
    if (rc != GS_OK) return throw_gs(env, rc);
    napi_value o, arr;
    NAPI_CALL(env, napi_create_object(env, &o));
    set_num(env, o, "numGaussians", (double)st.num_gaussians);
    set_num(env, o, "numVisible", (double)st.num_visible);
    set_num(env, o, "numIntersections", (double)st.num_intersections);
    set_num(env, o, "numProcessed", (double)st.num_processed);
    set_num(env, o, "numTiles", (double)st.num_tiles);
    set_num(env, o, "sortPasses", (double)st.sort_passes);
    set_num(env, o, "frames", (double)st.frames);
    set_num(env, o, "frameUs", (double)st.frame_us);
    set_num(env, o, "numEvaluated", (double)st.num_evaluated);
    set_num(env, o, "depthOrdered", (double)st.depth_ordered);
    set_num(env, o, "tightBinning", (double)st.tight_binning);
    set_num(env, o, "graphFrames", (double)st.graph_frames);
    set_num(env, o, "capacity", (double)st.capacity);
    set_num(env, o, "maxIntersectionsSeen", (double)st.max_intersections_seen);
    set_num(env, o, "truncatedFrames", (double)st.truncated_frames);
    NAPI_CALL(env, napi_create_array_with_length(env, GS_STAGE_COUNT, &arr));
    for (uint32_t i = 0; i < GS_STAGE_COUNT; ++i) {
        napi_value n;
        napi_create_double(env, (double)st.stage_us[i], &n);
        napi_set_element(env, arr, i, n);
    }
    napi_set_named_property(env, o, "stageUs", arr);
    return o;
}

static napi_value js_slab(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    gs_ctx* ctx = argc >= 1 ? unwrap(env, argv[0]) : NULL;
    if (!ctx) return NULL;
    uint32_t b = 0, w = 0;
    int32_t rc = gs_slab_width(ctx, &b, &w);
    if (rc != GS_OK) return throw_gs(env, rc);
    napi_value o;
    NAPI_CALL(env, napi_create_object(env, &o));
    set_num(env, o, "begin", b);
    set_num(env, o, "width", w);
    return o;
}

/* loadPly(path) -> {n, degree, records: ArrayBuffer}: the native PackedGaussians (gs_ply_load) */
static napi_value js_load_ply(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    char path[4096];
    size_t len = 0;
    if (argc < 1 || napi_get_value_string_utf8(env, argv[0], path, sizeof(path), &len) != napi_ok) {
        napi_throw_type_error(env, NULL, "gsplat.loadPly: path required");
        return NULL;
    }
    void* rec = NULL;
    uint64_t n = 0;
    int32_t degree = 0;
    int32_t rc = gs_ply_load(path, &rec, &n, &degree);
    if (rc != GS_OK) return throw_gs(env, rc);
    void* dst = NULL;
    napi_value ab, o;
    if (napi_create_arraybuffer(env, (size_t)n * GS_SPLAT_RECORD_BYTES, &dst, &ab) != napi_ok) {
        gs_ply_free(rec);
        napi_throw_error(env, NULL, "gsplat.loadPly: allocation failed");
        return NULL;
    }
    memcpy(dst, rec, (size_t)n * GS_SPLAT_RECORD_BYTES);
    gs_ply_free(rec);
    NAPI_CALL(env, napi_create_object(env, &o));
    set_num(env, o, "n", (double)n);
    set_num(env, o, "degree", (double)degree);
    napi_set_named_property(env, o, "records", ab);
    return o;
}

/* uploadPly(handle, path) -> n : the streaming loader (file -> pinned chunks -> device scene arrays, gs_upload_ply) */
static napi_value js_upload_ply(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    gs_ctx* ctx = argc >= 2 ? unwrap(env, argv[0]) : NULL;
    if (!ctx) return NULL;
    char path[4096];
    size_t len = 0;
    if (napi_get_value_string_utf8(env, argv[1], path, sizeof(path), &len) != napi_ok) {
        napi_throw_type_error(env, NULL, "gsplat.uploadPly: path required");
        return NULL;
    }
    uint64_t n = 0;
    int32_t rc = gs_upload_ply(ctx, path, &n);
    if (rc != GS_OK) return throw_gs(env, rc);
    napi_value v;
    NAPI_CALL(env, napi_create_double(env, (double)n, &v));
    return v;
}

/* hostAlloc(bytes) -> ArrayBuffer over page-locked memory (gs_host_alloc): a frame sink renderToSink copies into asynchronously */
static void finalize_pinned(napi_env env, void* data, void* hint) { (void)env; (void)hint; gs_host_free(data); }
static napi_value js_host_alloc(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    double bytes = 0;
    if (argc < 1 || napi_get_value_double(env, argv[0], &bytes) != napi_ok || !(bytes > 0.0) || bytes > 17179869184.0) {
        napi_throw_type_error(env, NULL, "gsplat.hostAlloc: byte count required");
        return NULL;
    }
    void* p = NULL;
    int32_t rc = gs_host_alloc((uint64_t)bytes, &p);
    if (rc != GS_OK) return throw_gs(env, rc);
    napi_value ab;
    if (napi_create_external_arraybuffer(env, p, (size_t)bytes, finalize_pinned, NULL, &ab) != napi_ok) {
        gs_host_free(p);
        napi_throw_error(env, NULL, "gsplat.hostAlloc: napi_create_external_arraybuffer failed");
        return NULL;
    }
    return ab;
}

typedef struct {
    napi_async_work work;
    napi_deferred deferred;
    ctx_box* box;
    gs_ctx* ctx;
    uint64_t ticket;
    napi_ref sink_ref; /* keeps the sink alive until the copy has landed */
    int32_t rc;
    char err[512];
} sink_job;

static void sink_execute(napi_env env, void* data) {
    (void)env;
    sink_job* j = (sink_job*)data;
    j->rc = gs_wait_ticket(j->ctx, j->ticket);
    if (j->rc != GS_OK) { strncpy(j->err, gs_last_error(), sizeof(j->err) - 1); j->err[sizeof(j->err) - 1] = 0; }
}

static void sink_complete(napi_env env, napi_status status, void* data) {
    sink_job* j = (sink_job*)data;
    ctx_box* box = j->box;
    napi_value v;
    box->inflight--;
    if (!box->inflight && !box->busy && box->destroy_pending) {
        if (box->ctx) gs_destroy(box->ctx);
        box->ctx = NULL;
        if (box->destroy_pending == 2) free(box);
        else box->destroy_pending = 0;
    }
    if (status == napi_ok && j->rc == GS_OK) {
        napi_get_undefined(env, &v);
        napi_resolve_deferred(env, j->deferred, v);
    } else {
        napi_value msg;
        napi_create_string_utf8(env, j->rc != GS_OK ? j->err : "gsplat: async work cancelled", NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, NULL, msg, &v);
        napi_reject_deferred(env, j->deferred, v);
    }
    napi_delete_reference(env, j->sink_ref);
    napi_delete_async_work(env, j->work);
    free(j);
}

/* renderToSink(handle, uniforms160, sink) -> Promise<void>: enqueues the frame and the copy of its pixels into `sink` NOW (on the
 * calling thread, without waiting: gs_render_host) and resolves when both are complete (gs_wait_ticket on a libuv worker).
 * Several may be outstanding: frame k+1 is enqueued while frame k is still being rendered and copied.  `sink`: ArrayBuffer or
 * view of at least height * slabWidth * 4 bytes, ideally from hostAlloc(). */
static napi_value js_render_to_sink(napi_env env, napi_callback_info info) {
    size_t argc = 3;
    napi_value argv[3];
    NAPI_CALL(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    ctx_box* box = argc >= 3 ? unbox(env, argv[0]) : NULL;
    if (!box) return NULL;
    if (!box->ctx || box->busy || box->destroy_pending) {
        napi_throw_error(env, NULL, "gsplat.renderToSink: the context is destroyed or owned by a renderAsync frame");
        return NULL;
    }
    void *udata = NULL, *sink = NULL;
    size_t ulen = 0, slen = 0;
    if (!get_bytes(env, argv[1], &udata, &ulen) || ulen < GS_UNIFORM_BYTES || !get_bytes(env, argv[2], &sink, &slen)) {
        napi_throw_type_error(env, NULL, "gsplat.renderToSink: need the 160-byte uniform block and a sink buffer");
        return NULL;
    }
    sink_job* j = (sink_job*)calloc(1, sizeof(sink_job));
    if (!j) { napi_throw_error(env, NULL, "gsplat.renderToSink: out of memory"); return NULL; }
    j->box = box;
    j->ctx = box->ctx;
    int32_t rc = gs_render_host(box->ctx, udata, sink, (uint64_t)slen, &j->ticket);
    if (rc != GS_OK) { free(j); return throw_gs(env, rc); }
    napi_value promise, name;
    if (napi_create_reference(env, argv[2], 1, &j->sink_ref) != napi_ok) {
        gs_wait_ticket(box->ctx, j->ticket); /* the copy must not outlive the buffer */
        free(j);
        napi_throw_error(env, NULL, "gsplat.renderToSink: N-API call failed");
        return NULL;
    }
    if (napi_create_promise(env, &j->deferred, &promise) != napi_ok ||
        napi_create_string_utf8(env, "gsplat.sink", NAPI_AUTO_LENGTH, &name) != napi_ok ||
        napi_create_async_work(env, NULL, name, sink_execute, sink_complete, j, &j->work) != napi_ok ||
        napi_queue_async_work(env, j->work) != napi_ok) {
        gs_wait_ticket(box->ctx, j->ticket);
        napi_delete_reference(env, j->sink_ref);
        free(j);
        napi_throw_error(env, NULL, "gsplat.renderToSink: N-API call failed");
        return NULL;
    }
    box->inflight++;
    return promise;
}

static napi_value init(napi_env env, napi_value exports) {
    static const struct { const char* name; napi_callback fn; } fns[] = {
        {"create", js_create},       {"destroy", js_destroy},         {"uploadSplats", js_upload},
        {"renderSync", js_render_sync}, {"renderAsync", js_render_async}, {"readRgba8", js_read_rgba8},
        {"readBuffer", js_read_buffer}, {"stats", js_stats},             {"slab", js_slab},
        {"loadPly", js_load_ply},    {"shareSplats", js_share},       {"uploadPly", js_upload_ply},
        {"hostAlloc", js_host_alloc}, {"renderToSink", js_render_to_sink},
    };
    for (size_t i = 0; i < sizeof(fns) / sizeof(fns[0]); ++i) {
        napi_value f;
        if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok) return NULL;
        napi_set_named_property(env, exports, fns[i].name, f);
    }
    set_num(env, exports, "abiVersion", (double)gs_abi_version());
    set_num(env, exports, "FLAG_EXACT_BLEND", GS_FLAG_EXACT_BLEND);
    set_num(env, exports, "FLAG_F32_TAP", GS_FLAG_F32_TAP);
    set_num(env, exports, "FLAG_TIMING", GS_FLAG_TIMING);
    return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
