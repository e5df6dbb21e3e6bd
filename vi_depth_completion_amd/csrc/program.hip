// Programs: a whole network (or the whole per-frame path) recorded once as an array of launch descriptors and executed
// by ONE native call -- eagerly (one launch per op, optional fork/join across up to VIDC_MAX_STREAMS HIP streams) or as
// a captured hipGraph.  This is what replaces the reference's ~1500 Python-dispatched ATen calls per frame
// (SURVEY.md §2.2): the Python host crosses the C ABI once per frame, not once per layer.
#include "common.h"
#include <vector>

struct vidc_program {
    std::vector<vidc_op> ops;
    hipStream_t side[VIDC_MAX_STREAMS] = {nullptr, nullptr, nullptr, nullptr};   // [0] unused: the caller's stream
    hipEvent_t ev[VIDC_MAX_STREAMS] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t fork_ev = nullptr;
    int n_streams = 1;
    hipGraph_t graph[VIDC_MAX_SEGMENTS] = {};      // [0] = the whole program (vidc_program_capture) or segment 0
    hipGraphExec_t exec[VIDC_MAX_SEGMENTS] = {};
    std::vector<hipEvent_t> op_ev;   // per-op timing events (eager timing mode)
};

namespace {

int launch_op(const vidc_op& op, hipStream_t st) {
    vidc_stream_t s = reinterpret_cast<vidc_stream_t>(st);
    const vidc_generic_args& g = op.u.g;
    switch (op.kind) {
        case VIDC_OP_CONV:
            return vidc_conv2d_bn_act(&op.u.conv, s);
        case VIDC_OP_STEM:
            if (g.p[4])      // input gathered through the forward warp: p[4] = warp parameter records, f = cx, cy, i[8] = align_corners
                return vidc_stem_conv3x3s2_warped((const float*)g.p[0], (const float*)g.p[4], (const float*)g.p[1], (float*)g.p[2], g.i[0], g.i[2], g.i[3], g.i[4],
                                                  g.i[5], g.i[6], const_cast<void*>(g.p[3]), g.i[7], g.f[0], g.f[1], g.i[8], s);
            return vidc_stem_conv3x3s2((const float*)g.p[0], (const float*)g.p[1], (float*)g.p[2], g.i[0], g.i[1], g.i[2], g.i[3],
                                       g.i[4], g.i[5], g.i[6], const_cast<void*>(g.p[3]), g.i[7], s);
        case VIDC_OP_MAXPOOL:
            return vidc_maxpool3x3s2((const float*)g.p[0], (float*)g.p[1], g.i[0], g.i[1], g.i[2], g.i[3], g.i[4], g.i[5],
                                     const_cast<void*>(g.p[2]), s);
        case VIDC_OP_UPSAMPLE:
            return vidc_upsample_bilinear_ac((const float*)g.p[0], (float*)g.p[1], g.i[0], g.i[1], g.i[2], g.i[3], g.i[4], g.i[5],
                                             g.i[6], g.i[7], g.i[8], const_cast<void*>(g.p[2]), s);
        case VIDC_OP_HEAD:
            return vidc_head_conv1x1_upsample((const float*)g.p[0], (const float*)g.p[1], (const float*)g.p[2], (float*)g.p[3],
                                              (float*)g.p[4], g.i[0], g.i[1], g.i[2], g.i[3], g.i[4], g.i[5], g.i[6], g.i[7],
                                              g.i[8], g.i[9], s);
        case VIDC_OP_WARP_PARAMS:
            return vidc_warp2dof_params((const float*)g.p[0], (const float*)g.p[1], g.i[0], g.f[0], g.f[1], g.f[2], g.f[3],
                                        (const float*)g.p[2], g.i[1], g.i[2], (float*)g.p[3], s);
        case VIDC_OP_WARP_FWD:
            return vidc_warp2dof_fwd((const float*)g.p[0], (const float*)g.p[1], (float*)g.p[2], g.i[0], g.i[1], g.i[2], g.i[3],
                                     g.f[0], g.f[1], g.i[4], s);
        case VIDC_OP_WARP_INV:
            return vidc_warp2dof_inv_rot_norm((const float*)g.p[0], (const float*)g.p[1], (float*)g.p[2], g.i[0], g.i[1], g.i[2],
                                              g.f[0], g.f[1], g.i[3], g.i[4], s);
        case VIDC_OP_SPLIT: {  // p[0] fp32 rows -> p[1] split-bf16 image; i[0..1] = rows (lo, hi), i[2] = C, i[3] = ldx
            long long rows = (long long)(uint32_t)g.i[0] | ((long long)(uint32_t)g.i[1] << 32);
            return vidc_split_bf16x3((const float*)g.p[0], const_cast<void*>(g.p[1]), rows, g.i[2], g.i[3], s);
        }
        case VIDC_OP_AVGPOOL:   // i = B, H, W, C, ldx, kh, kw, sh, sw, ph, pw, ldy
            return vidc_avgpool2d((const float*)g.p[0], (float*)g.p[1], g.i[0], g.i[1], g.i[2], g.i[3], g.i[4], g.i[5], g.i[6], g.i[7], g.i[8],
                                  g.i[9], g.i[10], g.i[11], s);
        case VIDC_OP_NORMALIZE:   // i = B, C, HW
            return vidc_normalize_nchw((const float*)g.p[0], (float*)g.p[1], g.i[0], g.i[1], g.i[2], s);
        case VIDC_OP_DET_IM2COL:   // i = B, H, W, Hp, Wp; f = mean (b, g, r)
            return vidc_det_stem_im2col((const float*)g.p[0], (float*)g.p[1], g.i[0], g.i[1], g.i[2], g.i[3], g.i[4], g.f[0], g.f[1], g.f[2], s);
        case VIDC_OP_NEAREST2X:    // i = B, h, w, C, ldx, ldy
            return vidc_upsample_nearest2x((const float*)g.p[0], (float*)g.p[1], g.i[0], g.i[1], g.i[2], g.i[3], g.i[4], g.i[5], s);
        case VIDC_OP_MASK:
            return vidc_mask_scale((const float*)g.p[0], (const float*)g.p[1], (float*)g.p[2], g.i[0], g.i[1], g.i[2], g.i[3], g.i[4], g.i[5], g.i[6], g.i[7], s);
        case VIDC_OP_WINO_IN:
            return vidc_winograd_input_transform((const float*)g.p[0], const_cast<void*>(g.p[1]), g.i[0], g.i[1], g.i[2], g.i[3], g.i[4], g.i[5], g.i[6],
                                                 g.i[7], g.i[8], s);
        case VIDC_OP_WINO_OUT:
            return vidc_winograd_output_transform((const float*)g.p[0], (float*)g.p[1], const_cast<void*>(g.p[2]), (const float*)g.p[3],
                                                  (const float*)g.p[4], (const float*)g.p[5], (const float*)g.p[6], g.i[0], g.i[1], g.i[2], g.i[3],
                                                  g.i[4], g.i[5], g.i[6], g.i[7], g.i[8], s);
        case VIDC_OP_COPY: {   // p[0] -> p[1], i[0..1] = byte count (lo, hi)
            size_t bytes = (size_t)(uint32_t)g.i[0] | ((size_t)(uint32_t)g.i[1] << 32);
            VIDC_HIP(hipMemcpyAsync(const_cast<void*>(g.p[1]), g.p[0], bytes, hipMemcpyDeviceToDevice, st));
            return VIDC_OK;
        }
        default:
            vidc::set_error("program: unknown op kind %d", op.kind);
            return VIDC_ERR_SHAPE;
    }
}

// Issues every op; ops with stream_id k>0 go to the program's side stream k.  Dependencies across stream ids are
// expressed by wait_mask (join on everything issued so far on those ids).  Works identically under stream capture,
// where the event record/wait pairs become graph edges.
int issue(vidc_program* p, hipStream_t main, bool timing, size_t begin = 0, size_t end = (size_t)-1) {
    if (end > p->ops.size()) end = p->ops.size();
    hipStream_t st[VIDC_MAX_STREAMS];
    st[0] = main;
    for (int k = 1; k < VIDC_MAX_STREAMS; ++k) st[k] = p->side[k];
    bool forked[VIDC_MAX_STREAMS] = {true, false, false, false};
    bool dirty[VIDC_MAX_STREAMS] = {false, false, false, false};
    for (size_t i = begin; i < end; ++i) {
        const vidc_op& op = p->ops[i];
        const int sid = op.stream_id;
        if (!forked[sid]) {   // side stream joins the main stream's history on first use
            VIDC_HIP(hipEventRecord(p->fork_ev, main));
            VIDC_HIP(hipStreamWaitEvent(st[sid], p->fork_ev, 0));
            forked[sid] = true;
        }
        for (int k = 0; k < p->n_streams; ++k)
            if (k != sid && (op.wait_mask >> k & 1)) {
                VIDC_HIP(hipEventRecord(p->ev[k], st[k]));
                VIDC_HIP(hipStreamWaitEvent(st[sid], p->ev[k], 0));
                if (sid == 0) dirty[k] = false;
            }
        if (timing) VIDC_HIP(hipEventRecord(p->op_ev[i], st[sid]));
        int rc = launch_op(op, st[sid]);
        if (rc != VIDC_OK) return rc;
        dirty[sid] = true;
    }
    if (timing) VIDC_HIP(hipEventRecord(p->op_ev[end], main));
    for (int k = 1; k < p->n_streams; ++k)   // final join: the caller's stream owns everything afterwards
        if (forked[k] && dirty[k]) {
            VIDC_HIP(hipEventRecord(p->ev[k], st[k]));
            VIDC_HIP(hipStreamWaitEvent(main, p->ev[k], 0));
        }
    return VIDC_OK;
}

}  // namespace

extern "C" int vidc_program_create(const vidc_op* ops, int n_ops, vidc_program** out) {
    VIDC_REQUIRE(ops && out, VIDC_ERR_NULL, "vidc_program_create: null pointer");
    VIDC_REQUIRE(n_ops > 0, VIDC_ERR_SHAPE, "vidc_program_create: empty program");
    vidc_program* p = new vidc_program();
    p->ops.assign(ops, ops + n_ops);
    for (const vidc_op& op : p->ops) {
        if (op.stream_id < 0 || op.stream_id >= VIDC_MAX_STREAMS) {
            delete p;
            vidc::set_error("vidc_program_create: stream_id %d out of range", op.stream_id);
            return VIDC_ERR_SHAPE;
        }
        if (op.stream_id + 1 > p->n_streams) p->n_streams = op.stream_id + 1;
    }
    hipError_t e = hipSuccess;
    for (int k = 1; k < p->n_streams && e == hipSuccess; ++k) e = hipStreamCreateWithFlags(&p->side[k], hipStreamNonBlocking);
    for (int k = 0; k < p->n_streams && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&p->ev[k], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->fork_ev, hipEventDisableTiming);
    if (e != hipSuccess) {
        vidc::set_error("vidc_program_create: stream/event creation failed: %s", hipGetErrorString(e));
        vidc_program_destroy(p);      // frees whatever was created
        return VIDC_ERR_HIP;
    }
    *out = p;
    return VIDC_OK;
}

extern "C" int vidc_program_run(vidc_program* p, vidc_stream_t stream) {
    VIDC_REQUIRE(p, VIDC_ERR_STATE, "vidc_program_run: null program");
    return issue(p, vidc::as_stream(stream), false);
}

namespace {
int capture_range(vidc_program* p, hipStream_t st, size_t begin, size_t end, int seg) {
    if (p->exec[seg]) { hipGraphExecDestroy(p->exec[seg]); p->exec[seg] = nullptr; }
    if (p->graph[seg]) { hipGraphDestroy(p->graph[seg]); p->graph[seg] = nullptr; }
    VIDC_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    int rc = issue(p, st, false, begin, end);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(st, &g);
    if (rc != VIDC_OK) { if (g) hipGraphDestroy(g); return rc; }
    if (e != hipSuccess) { vidc::set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e)); return VIDC_ERR_HIP; }
    p->graph[seg] = g;
    VIDC_HIP(hipGraphInstantiate(&p->exec[seg], p->graph[seg], nullptr, nullptr, 0));
    return VIDC_OK;
}
}  // namespace

extern "C" int vidc_program_capture(vidc_program* p, vidc_stream_t stream) {
    VIDC_REQUIRE(p, VIDC_ERR_STATE, "vidc_program_capture: null program");
    hipStream_t st = vidc::as_stream(stream);
    VIDC_REQUIRE(st != nullptr, VIDC_ERR_STATE, "vidc_program_capture: needs a non-default stream");
    return capture_range(p, st, 0, p->ops.size(), 0);
}

extern "C" int vidc_program_launch(vidc_program* p, vidc_stream_t stream) {
    VIDC_REQUIRE(p && p->exec[0], VIDC_ERR_STATE, "vidc_program_launch: program not captured");
    for (int k = 0; k < VIDC_MAX_SEGMENTS && p->exec[k]; ++k)      // a program captured in segments replays all of them, in order
        VIDC_HIP(hipGraphLaunch(p->exec[k], vidc::as_stream(stream)));
    return VIDC_OK;
}

// Segments: ops [begin, end) of a program as their own eager run / hipGraph, so the host can interleave other work
// (the plane block and its device->host read) between parts of one planned program.
extern "C" int vidc_program_run_range(vidc_program* p, vidc_stream_t stream, int begin, int end) {
    VIDC_REQUIRE(p, VIDC_ERR_STATE, "vidc_program_run_range: null program");
    VIDC_REQUIRE(begin >= 0 && begin < end && (size_t)end <= p->ops.size(), VIDC_ERR_SHAPE, "vidc_program_run_range: bad range [%d, %d)", begin, end);
    return issue(p, vidc::as_stream(stream), false, (size_t)begin, (size_t)end);
}

extern "C" int vidc_program_capture_range(vidc_program* p, vidc_stream_t stream, int begin, int end, int segment) {
    VIDC_REQUIRE(p, VIDC_ERR_STATE, "vidc_program_capture_range: null program");
    VIDC_REQUIRE(begin >= 0 && begin < end && (size_t)end <= p->ops.size(), VIDC_ERR_SHAPE, "vidc_program_capture_range: bad range [%d, %d)", begin, end);
    VIDC_REQUIRE(segment >= 0 && segment < VIDC_MAX_SEGMENTS, VIDC_ERR_SHAPE, "vidc_program_capture_range: segment %d out of range", segment);
    hipStream_t st = vidc::as_stream(stream);
    VIDC_REQUIRE(st != nullptr, VIDC_ERR_STATE, "vidc_program_capture_range: needs a non-default stream");
    return capture_range(p, st, (size_t)begin, (size_t)end, segment);
}

extern "C" int vidc_program_launch_segment(vidc_program* p, vidc_stream_t stream, int segment) {
    VIDC_REQUIRE(p && segment >= 0 && segment < VIDC_MAX_SEGMENTS && p->exec[segment], VIDC_ERR_STATE,
                 "vidc_program_launch_segment: segment not captured");
    VIDC_HIP(hipGraphLaunch(p->exec[segment], vidc::as_stream(stream)));
    return VIDC_OK;
}

extern "C" int vidc_program_time(vidc_program* p, vidc_stream_t stream, int iters, int use_graph, float* ms_out, float* per_op_ms) {
    VIDC_REQUIRE(p && ms_out, VIDC_ERR_NULL, "vidc_program_time: null pointer");
    VIDC_REQUIRE(iters > 0, VIDC_ERR_SHAPE, "vidc_program_time: iters must be > 0");
    hipStream_t st = vidc::as_stream(stream);
    hipEvent_t t0, t1;
    VIDC_HIP(hipEventCreate(&t0));
    VIDC_HIP(hipEventCreate(&t1));
    int rc = VIDC_OK;
    if (per_op_ms && !use_graph) {
        const size_t n = p->ops.size();
        if (p->op_ev.size() != n + 1) {
            p->op_ev.resize(n + 1);
            for (auto& e : p->op_ev) VIDC_HIP(hipEventCreate(&e));
        }
        for (size_t i = 0; i < n; ++i) per_op_ms[i] = 0.f;
        float total = 0.f;
        for (int it = 0; it < iters && rc == VIDC_OK; ++it) {
            rc = issue(p, st, true);
            if (rc != VIDC_OK) break;
            VIDC_HIP(hipStreamSynchronize(st));
            for (size_t i = 0; i < n; ++i) {   // start-to-next-start on the issuing order (exact when single-stream)
                float ms = 0.f;
                hipEventElapsedTime(&ms, p->op_ev[i], p->op_ev[i + 1]);
                per_op_ms[i] += ms / iters;
            }
            float ms = 0.f;
            hipEventElapsedTime(&ms, p->op_ev[0], p->op_ev[n]);
            total += ms;
        }
        ms_out[0] = total / iters;
    } else {
        VIDC_HIP(hipEventRecord(t0, st));
        for (int it = 0; it < iters && rc == VIDC_OK; ++it)
            rc = use_graph ? vidc_program_launch(p, stream) : issue(p, st, false);
        VIDC_HIP(hipEventRecord(t1, st));
        VIDC_HIP(hipEventSynchronize(t1));
        float ms = 0.f;
        VIDC_HIP(hipEventElapsedTime(&ms, t0, t1));
        ms_out[0] = ms / iters;
    }
    hipEventDestroy(t0);
    hipEventDestroy(t1);
    return rc;
}

extern "C" int vidc_program_destroy(vidc_program* p) {
    if (!p) return VIDC_OK;
    for (int k = 0; k < VIDC_MAX_SEGMENTS; ++k) {
        if (p->exec[k]) hipGraphExecDestroy(p->exec[k]);
        if (p->graph[k]) hipGraphDestroy(p->graph[k]);
    }
    for (int k = 1; k < VIDC_MAX_STREAMS; ++k) if (p->side[k]) hipStreamDestroy(p->side[k]);
    for (int k = 0; k < VIDC_MAX_STREAMS; ++k) if (p->ev[k]) hipEventDestroy(p->ev[k]);
    if (p->fork_ev) hipEventDestroy(p->fork_ev);
    for (auto& e : p->op_ev) hipEventDestroy(e);
    delete p;
    return VIDC_OK;
}
