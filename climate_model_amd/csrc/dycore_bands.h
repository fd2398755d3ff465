// dycore_bands.h -- the latitude-band Matsuno step inside the library (included by dycore_hip.hip).
//
// New functionality (the reference is single-device; its only "exchange" is misc_boundaries.py:22-82).
// One band per GPU / process.  Per substep a band refreshes its north / south halo rows from its neighbours twice,
// each time behind a kernel that does not need the rows:
//
//   main:  |B1  continuity ---- momentum inner tiles - |AB2 - momentum edge tiles ---- temperature(+moisture)
//   comm:                  AB2: pack, transport, unpack                              B1: pack, transport, unpack
// and, on a handle whose momentum and temperature kernels are one launch (the whole-substep kernel, round 3):
//   main:  |B1  continuity ---- first half of the inner tiles - |AB2 - edge tiles ---- second half of the inner tiles
//   comm:                  AB2: pack, transport, unpack                             B1: pack, transport, unpack
//
//   AB2 = 2-D column pressures at the u- / v-points, WWIND, COLP_NEW of this substep + the POTT (QV, QC) rows of the
//         state it evaluates on and the rows of that state's diagnostics as their owner computed them (PVTFVB and the
//         PHIVB planes the momentum kernel's level chunks start from; dycore_band_row_fields);
//         B1 = the new winds;  B2 = the scalar rows of the last substep of a call.
//
// The schedule, the message plans, the pack / unpack launches and the stream / event choreography of one step are
// issued by ONE call (dycore_bands_step_matsuno): the host cost of a band's step is a few dozen HIP calls.
// Transport: RCCL point-to-point (ncclSend / ncclRecv inside one group per exchange, on the comm stream; librccl
// is resolved at run time so that a single-GPU host needs none), or device-to-device copies between the bands of
// one process (the single-GPU rehearsal of the same schedule, and its bit-for-bit test).
#pragma once
#include <dlfcn.h>
// Every RCCL entry point is resolved with dlsym, so only a handful of types are needed at compile time: the RCCL
// development header when the ROCm install has it, otherwise the same declarations spelt out (the NCCL ABI: an opaque
// communicator, a 128-byte id, result 0 = success, the datatype codes) -- a ROCm without rccl-dev still builds the library.
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5,
               ncclFloat16 = 6, ncclHalf = 6, ncclFloat32 = 7, ncclFloat = 7, ncclFloat64 = 8, ncclDouble = 8 } ncclDataType_t;
#endif

namespace {

// message elements have the working precision of the build (counts and buffers are in `real`)
constexpr ncclDataType_t kNcclReal = sizeof(real) == 4 ? ncclFloat : ncclDouble;

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

// librccl of the process: the copy that is already mapped (a PyTorch-ROCm host brings its own, bound to the same
// HIP runtime this library runs on), else the system one
int load_rccl(RcclApi **out)
{
    static RcclApi api;
    static int state = 0;       // 0 not tried, 1 ok, -1 failed
    if (state == 0) {
        const char *names[] = {"librccl.so.1", "librccl.so"};      // the SONAME first: RTLD_NOLOAD matches a loaded copy by it
        for (const char *n : names)
            if ((api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!api.lib)
            for (const char *n : names)
                if ((api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        state = -1;
        if (api.lib) {
            api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
            api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
            api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
            api.Send = (decltype(api.Send))dlsym(api.lib, "ncclSend");
            api.Recv = (decltype(api.Recv))dlsym(api.lib, "ncclRecv");
            api.GroupStart = (decltype(api.GroupStart))dlsym(api.lib, "ncclGroupStart");
            api.GroupEnd = (decltype(api.GroupEnd))dlsym(api.lib, "ncclGroupEnd");
            api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
            if (api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.Send && api.Recv && api.GroupStart && api.GroupEnd)
                state = 1;
        }
    }
    if (state != 1) return fail(DYCORE_ESTATE, "librccl.so could not be loaded (or lacks ncclSend / ncclRecv)");
    *out = &api;
    return 0;
}

enum { PH_AB2 = 0, PH_B1 = 1, PH_B2 = 2, PH_COUNT = 3 };

// one row of one field inside a message; `est` != nullptr marks a state field (UWIND, VWIND, POTT, QV, QC), which the
// fused step keeps in its *_EST scratch buffer after the predictor
struct MsgItem { real *Ptrs::*plain; real *Ptrs::*est; int row, nk, k0 = 0, kstep = 1; };

struct Msg {
    std::vector<MsgItem> send_s, send_n, recv_s, recv_n;
    size_t ss = 0, sn = 0, rs = 0, rn = 0;      // elements (`real`)
    real *send = nullptr, *recv = nullptr;    // [south part | north part]
};

struct BandMember {
    dycore *h = nullptr;
    bool has_s = false, has_n = false;
    Msg msg[PH_COUNT];
};

}  // namespace

struct dycore_bands {
    std::vector<BandMember> m;
    int kind = 0;                               // 0: bands of one process (device copies), 1: RCCL
    hipStream_t main = nullptr, comm = nullptr;
    hipEvent_t ev_main = nullptr;
    hipEvent_t done[PH_COUNT] = {nullptr, nullptr, nullptr};
    bool pending[PH_COUNT] = {false, false, false};
    RcclApi *rccl = nullptr;
    ncclComm_t nccl = nullptr;
    int rank = 0, nranks = 1, south = -1, north = -1;
    std::vector<real *> allocs;
};

namespace {

size_t items_size(const dycore *h, const std::vector<MsgItem> &v)
{
    size_t n = 0;
    for (const MsgItem &it : v) n += (size_t)it.nk * (h->g.nx + 3);
    return n;
}

// BandHalo of bands.py for a handle whose fused step re-forms the mass fluxes from the winds
void build_plan(BandMember &b)
{
    const dycore *h = b.h;
    const int N = h->g.ny, nz = h->g.nz;
    const bool moist = h->g.i_moist;
#define ST(n_) &Ptrs::n_, &Ptrs::n_##_EST
    // the scalar state and, with it, the diagnostics of that state the owner computed (dycore_band_row_fields): a band
    // does not re-diagnose the rows it receives
    std::vector<const char *> diag;
    band_row_fields(h, diag);
    auto scalars = [&](std::vector<MsgItem> &v, int row) {
        v.push_back({ST(POTT), row, nz});
        if (moist) { v.push_back({ST(QV), row, nz}); v.push_back({ST(QC), row, nz}); }
        const bool planes = !h->g.i_turb && phi_on_demand(h);
        for (const char *n : diag) {
            const FieldDesc &d = kFields[h->field_idx.at(n)];
            if (planes && d.member == &Ptrs::PHIVB) {       // the planes PHIVB_STRIDE, 2 PHIVB_STRIDE, ... below the surface interface
                const int np = (nz - 1) / PHIVB_STRIDE;
                if (np > 0) { MsgItem it{d.member, nullptr, row, np}; it.k0 = PHIVB_STRIDE; it.kstep = PHIVB_STRIDE; v.push_back(it); }
            } else
                v.push_back({d.member, nullptr, row, h->nk_of(d.nk)});
        }
    };
    {   // AB2
        Msg &m = b.msg[PH_AB2];
        m.send_n = {{&Ptrs::UCOL, nullptr, N, 1}, {&Ptrs::VCOL, nullptr, N, 1}, {&Ptrs::WWIND, nullptr, N, nz + 1}, {&Ptrs::COLP_NEW, nullptr, N, 1}};
        scalars(m.send_n, N);
        m.send_s = {{&Ptrs::UCOL, nullptr, 1, 1}, {&Ptrs::VCOL, nullptr, 1, 1}, {&Ptrs::VCOL, nullptr, 2, 1}, {&Ptrs::WWIND, nullptr, 1, nz + 1}, {&Ptrs::COLP_NEW, nullptr, 1, 1}};
        scalars(m.send_s, 1);
        m.recv_s = {{&Ptrs::UCOL, nullptr, 0, 1}, {&Ptrs::VCOL, nullptr, 0, 1}, {&Ptrs::WWIND, nullptr, 0, nz + 1}, {&Ptrs::COLP_NEW, nullptr, 0, 1}};
        scalars(m.recv_s, 0);
        m.recv_n = {{&Ptrs::UCOL, nullptr, N + 1, 1}, {&Ptrs::VCOL, nullptr, N + 1, 1}, {&Ptrs::VCOL, nullptr, N + 2, 1}, {&Ptrs::WWIND, nullptr, N + 1, nz + 1}, {&Ptrs::COLP_NEW, nullptr, N + 1, 1}};
        scalars(m.recv_n, N + 1);
    }
    {   // B1: the new winds; the second v-row beyond the northern edge travels with them
        Msg &m = b.msg[PH_B1];
        m.send_n = {{ST(UWIND), N, nz}, {ST(VWIND), N, nz}};
        m.send_s = {{ST(UWIND), 1, nz}, {ST(VWIND), 1, nz}, {ST(VWIND), 2, nz}};
        m.recv_s = {{ST(UWIND), 0, nz}, {ST(VWIND), 0, nz}};
        m.recv_n = {{ST(UWIND), N + 1, nz}, {ST(VWIND), N + 1, nz}, {ST(VWIND), N + 2, nz}};
    }
    {   // B2
        Msg &m = b.msg[PH_B2];
        scalars(m.send_n, N); scalars(m.send_s, 1); scalars(m.recv_s, 0); scalars(m.recv_n, N + 1);
    }
#undef ST
    for (int ph = 0; ph < PH_COUNT; ph++) {
        Msg &m = b.msg[ph];
        m.ss = b.has_s ? items_size(h, m.send_s) : 0;
        m.sn = b.has_n ? items_size(h, m.send_n) : 0;
        m.rs = b.has_s ? items_size(h, m.recv_s) : 0;
        m.rn = b.has_n ? items_size(h, m.recv_n) : 0;
    }
}

// pack (unpack = 0) / unpack (1) the south + north lists of one message on the handle's current stream.
// `src` = the substep whose output the state fields are taken from (0: the *_EST buffers)
int rows_move(dycore *h, const std::vector<MsgItem> &south, bool with_s, const std::vector<MsgItem> &north, bool with_n,
              int src, real *buf, int unpack)
{
    RowList rl;
    rl.n = 0;
    long long off = 0;
    int maxk = 1;
    auto flush = [&]() -> int {
        if (!rl.n) return 0;
        Scope s(h, K_HALO);
        hipLaunchKernelGGL(k_rows, dim3((h->g.nx + 3 + 255) / 256, maxk, rl.n), dim3(256), 0, h->stream, h->g, rl, buf, unpack);
        rl.n = 0; maxk = 1;
        return launch_check("k_rows");
    };
    for (int part = 0; part < 2; part++) {
        if (part == 0 ? !with_s : !with_n) continue;
        for (const MsgItem &it : (part == 0 ? south : north)) {
            const int e = rl.n++;
            rl.base[e] = h->p.*((src == 0 && it.est) ? it.est : it.plain);
            rl.row[e] = it.row; rl.nk[e] = it.nk; rl.k0[e] = it.k0; rl.kstep[e] = it.kstep; rl.off[e] = off;
            off += (long long)it.nk * (h->g.nx + 3);
            if (it.nk > maxk) maxk = it.nk;
            if (rl.n == ROWLIST_MAX) { int rc = flush(); if (rc) return rc; }
        }
    }
    return flush();
}

#define NCCL_TRY(B, expr)                                                                                         \
    do {                                                                                                          \
        ncclResult_t r_ = (expr);                                                                                 \
        if (r_ != ncclSuccess)                                                                                    \
            return fail(DYCORE_EHIP, "%s failed: %s", #expr, (B)->rccl->GetErrorString ? (B)->rccl->GetErrorString(r_) : "rccl error"); \
    } while (0)

// transport of the packed buffers on the comm stream
int bands_transport(dycore_bands *B, int ph)
{
    if (B->kind == 1) {
        Msg &m = B->m[0].msg[ph];
        // order inside the group: to the south, from the north, to the north, from the south -- any order is right
        // between distinct peers; it also pairs the messages of a band that is its own neighbour on both sides
        NCCL_TRY(B, B->rccl->GroupStart());
        if (B->south >= 0) NCCL_TRY(B, B->rccl->Send(m.send, m.ss, kNcclReal, B->south, B->nccl, B->comm));
        if (B->north >= 0) NCCL_TRY(B, B->rccl->Recv(m.recv + m.rs, m.rn, kNcclReal, B->north, B->nccl, B->comm));
        if (B->north >= 0) NCCL_TRY(B, B->rccl->Send(m.send + m.ss, m.sn, kNcclReal, B->north, B->nccl, B->comm));
        if (B->south >= 0) NCCL_TRY(B, B->rccl->Recv(m.recv, m.rs, kNcclReal, B->south, B->nccl, B->comm));
        NCCL_TRY(B, B->rccl->GroupEnd());
        return 0;
    }
    const int n = (int)B->m.size();
    for (int r = 0; r < n; r++) {
        Msg &m = B->m[r].msg[ph];
        if (r > 0) {            // from the southern neighbour: its north part
            Msg &o = B->m[r - 1].msg[ph];
            HIP_TRY(hipMemcpyAsync(m.recv, o.send + o.ss, m.rs * sizeof(real), hipMemcpyDeviceToDevice, B->comm));
        }
        if (r < n - 1) {        // from the northern neighbour: its south part
            Msg &o = B->m[r + 1].msg[ph];
            HIP_TRY(hipMemcpyAsync(m.recv + m.rs, o.send, m.rn * sizeof(real), hipMemcpyDeviceToDevice, B->comm));
        }
    }
    return 0;
}

// on the comm stream: pack, transport, unpack; one event per phase marks "halo rows of this phase are in place"
int bands_exchange(dycore_bands *B, int ph, int sub, real dt)
{
    int rc = 0;
    const int src = ph == PH_AB2 ? 1 - sub : sub;
    // everything of an exchange runs on the comm stream, the pack included: the rows it reads are complete when the
    // event below is recorded, and no kernel the main stream issues before it waits for this exchange's successor
    // writes them -- the main stream goes straight on to the kernel that hides the exchange
    HIP_TRY(hipEventRecord(B->ev_main, B->main));
    HIP_TRY(hipStreamWaitEvent(B->comm, B->ev_main, 0));
    for (BandMember &b : B->m) b.h->stream = B->comm;
    for (BandMember &b : B->m)
        if ((rc = rows_move(b.h, b.msg[ph].send_s, b.has_s, b.msg[ph].send_n, b.has_n, src, b.msg[ph].send, 0))) break;
    if (!rc) rc = bands_transport(B, ph);
    if (rc) {
        for (BandMember &b : B->m) b.h->stream = b.h->home;
        return rc;
    }
    for (BandMember &b : B->m)
        if ((rc = rows_move(b.h, b.msg[ph].recv_s, b.has_s, b.msg[ph].recv_n, b.has_n, src, b.msg[ph].recv, 1))) break;
    for (BandMember &b : B->m) b.h->stream = b.h->home;
    if (rc) return rc;
    HIP_TRY(hipEventRecord(B->done[ph], B->comm));
    B->pending[ph] = true;
    return 0;
}

int bands_wait(dycore_bands *B, int ph)
{
    if (!B->pending[ph]) return 0;
    HIP_TRY(hipStreamWaitEvent(B->main, B->done[ph], 0));
    B->pending[ph] = false;
    return 0;
}

int bands_finish_create(dycore_bands *B)
{
    // every member: the fused step with split parts and fluxes re-formed from the winds (the message plan above)
    for (BandMember &b : B->m) {
        if (!fused_path(b.h) || !flux_from_winds(b.h))
            return fail(DYCORE_ESTATE, "the in-library band step needs the fused step (dycore_overlap_parts, dycore_flux_from_winds): %s",
                        !b.h->zonal ? "lat_rad / lat_is_rad / corf / corf_is vary along a row, so this handle steps on the stage kernels"
                        : !b.h->fused ? "DYCORE_FUSED=0" : "the level count or a DYCORE_* switch selects kernels that store the mass fluxes");
        if (b.h->home != B->main) return fail(DYCORE_EINVAL, "all bands of one process must launch on the same stream");
        build_plan(b);
    }
    HIP_TRY(hipSetDevice(B->m[0].h->cfg.device));
    HIP_TRY(hipStreamCreateWithFlags(&B->comm, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&B->ev_main, hipEventDisableTiming));
    for (int ph = 0; ph < PH_COUNT; ph++) HIP_TRY(hipEventCreateWithFlags(&B->done[ph], hipEventDisableTiming));
    for (BandMember &b : B->m)
        for (int ph = 0; ph < PH_COUNT; ph++) {
            Msg &m = b.msg[ph];
            real *d = nullptr;
            HIP_TRY(hipMalloc((void **)&d, (m.ss + m.sn + 1) * sizeof(real)));
            B->allocs.push_back(d); m.send = d;
            HIP_TRY(hipMalloc((void **)&d, (m.rs + m.rn + 1) * sizeof(real)));
            B->allocs.push_back(d); m.recv = d;
        }
    return 0;
}

}  // namespace

extern "C" {

int dycore_bands_create_local(dycore_t *const *handles, int32_t n, dycore_bands_t **out)
{
    if (!handles || n < 1 || !out) return fail(DYCORE_EINVAL, "bad argument");
    dycore_bands *B = new dycore_bands();
    B->kind = 0;
    int joff = 0;
    for (int r = 0; r < n; r++) {
        dycore *h = handles[r];
        if (!h || h->g.joff != joff || h->g.nx != handles[0]->g.nx || h->g.nz != handles[0]->g.nz || h->g.ny_global != handles[0]->g.ny_global) {
            delete B;
            return fail(DYCORE_EINVAL, "band %d does not continue the grid at global row %d", r, joff + 1);
        }
        joff += h->g.ny;
        BandMember b;
        b.h = h; b.has_s = r > 0; b.has_n = r < n - 1;
        B->m.push_back(b);
    }
    if (joff != handles[0]->g.ny_global) { delete B; return fail(DYCORE_EINVAL, "the bands cover %d of %d rows", joff, handles[0]->g.ny_global); }
    B->main = handles[0]->home;
    int rc = bands_finish_create(B);
    if (rc) { dycore_bands_destroy(B); return rc; }
    *out = B;
    return 0;
}

int dycore_bands_unique_id(void *id, int32_t nbytes)
{
    if (!id || nbytes != (int32_t)sizeof(ncclUniqueId)) return fail(DYCORE_EINVAL, "the id buffer must hold %d bytes", (int)sizeof(ncclUniqueId));
    RcclApi *api;
    int rc = load_rccl(&api);
    if (rc) return rc;
    ncclUniqueId u;
    ncclResult_t r = api->GetUniqueId(&u);
    if (r != ncclSuccess) return fail(DYCORE_EHIP, "ncclGetUniqueId failed: %s", api->GetErrorString ? api->GetErrorString(r) : "rccl error");
    memcpy(id, &u, sizeof u);
    return 0;
}

int dycore_bands_create_rccl(dycore_t *h, const void *id, int32_t nbytes, int32_t rank, int32_t nranks,
                             int32_t south_rank, int32_t north_rank, dycore_bands_t **out)
{
    if (!h || !id || !out || nbytes != (int32_t)sizeof(ncclUniqueId) || rank < 0 || rank >= nranks ||
        south_rank >= nranks || north_rank >= nranks)
        return fail(DYCORE_EINVAL, "bad argument");
    if ((south_rank >= 0) == (h->g.pole_s != 0) || (north_rank >= 0) == (h->g.pole_n != 0))
        return fail(DYCORE_EINVAL, "a band has a neighbour exactly on the sides where it does not touch a pole");
    dycore_bands *B = new dycore_bands();
    B->kind = 1;
    B->rank = rank; B->nranks = nranks; B->south = south_rank; B->north = north_rank;
    BandMember b;
    b.h = h; b.has_s = south_rank >= 0; b.has_n = north_rank >= 0;
    B->m.push_back(b);
    B->main = h->home;
    int rc = load_rccl(&B->rccl);
    if (!rc) rc = bands_finish_create(B);
    if (!rc) {
        ncclUniqueId u;
        memcpy(&u, id, sizeof u);
        ncclResult_t r = B->rccl->CommInitRank(&B->nccl, nranks, u, rank);
        if (r != ncclSuccess) rc = fail(DYCORE_EHIP, "ncclCommInitRank failed: %s", B->rccl->GetErrorString ? B->rccl->GetErrorString(r) : "rccl error");
    }
    if (rc) { dycore_bands_destroy(B); return rc; }
    *out = B;
    return 0;
}

int dycore_bands_destroy(dycore_bands_t *B)
{
    if (!B) return 0;
    if (B->comm) (void)hipStreamSynchronize(B->comm);
    if (B->main) (void)hipStreamSynchronize(B->main);
    if (B->nccl && B->rccl) (void)B->rccl->CommDestroy(B->nccl);
    for (real *d : B->allocs) (void)hipFree(d);
    if (B->ev_main) (void)hipEventDestroy(B->ev_main);
    for (int ph = 0; ph < PH_COUNT; ph++)
        if (B->done[ph]) (void)hipEventDestroy(B->done[ph]);
    if (B->comm) (void)hipStreamDestroy(B->comm);
    delete B;
    return 0;
}

// One transport of the (still meaningless) AB2 send buffers, nothing unpacked: RCCL sets a peer connection up at
// the first send / recv between two ranks, which belongs in front of a timed region.  Every rank calls it.
int dycore_bands_connect(dycore_bands_t *B)
{
    if (!B) return fail(DYCORE_EINVAL, "null argument");
    HIP_TRY(hipEventRecord(B->ev_main, B->main));
    HIP_TRY(hipStreamWaitEvent(B->comm, B->ev_main, 0));
    int rc = bands_transport(B, PH_AB2);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(B->comm));
    return 0;
}

int dycore_bands_step_matsuno(dycore_bands_t *B, double dt, int32_t nsteps, int32_t with_rho)
{
    if (!B || nsteps < 0) return fail(DYCORE_EINVAL, "bad argument");
    int rc;
    for (BandMember &b : B->m)
        if ((rc = check_ready(b.h))) return rc;
    if (B->m.size() == 1 && !B->m[0].has_s && !B->m[0].has_n) {      // a single band is the whole grid
        for (int s = 0; s < nsteps; s++)
            if ((rc = do_step(B->m[0].h, dt, with_rho))) return rc;
        return 0;
    }
#define ALL(call) for (BandMember &b : B->m) { dycore *h = b.h; if ((rc = (call))) return rc; }
    // a turbulence-coupled step re-diagnoses RHO / RHOVB of the whole band, halo rows included, when it begins: the POTT
    // rows of the state it starts from go out alone at the end of every step and must have arrived by then
    const bool turb = B->m[0].h->g.i_turb != 0;
    for (int s = 0; s < nsteps; s++) {
        if (turb && (rc = bands_wait(B, PH_B2))) return rc;
        ALL(step_begin(h, with_rho))
        for (int sub = 0; sub < 2; sub++) {
            if ((rc = bands_wait(B, PH_B1)) || (rc = bands_wait(B, PH_B2))) return rc;
            ALL(substep_part(h, dt, sub, 0))
            if ((rc = bands_exchange(B, PH_AB2, sub, dt))) return rc;
            // (whole-substep kernel: INNER_A / INNER_B are the two halves of the inner tile rows, part TEMP holds no kernel;
            // two kernels: INNER_A is the whole inner piece, INNER_B nothing, part TEMP the temperature kernel)
            ALL(substep_part(h, dt, sub, DYCORE_PART_MOM_INNER_A))
            if ((rc = bands_wait(B, PH_AB2))) return rc;
            ALL(substep_part(h, dt, sub, DYCORE_PART_MOM_EDGE))
            if ((rc = bands_exchange(B, PH_B1, sub, dt))) return rc;
            ALL(substep_part(h, dt, sub, DYCORE_PART_MOM_INNER_B))
            ALL(substep_part(h, dt, sub, DYCORE_PART_TEMP))
        }
        if (turb && s < nsteps - 1 && (rc = bands_exchange(B, PH_B2, 1, dt))) return rc;
    }
#undef ALL
    if (nsteps > 0) {       // leave with the halo rows of the final state in place and every stream joined
        if ((rc = bands_exchange(B, PH_B2, 1, dt))) return rc;
    }
    if ((rc = bands_wait(B, PH_B1)) || (rc = bands_wait(B, PH_B2))) return rc;
    return 0;
}

}  // extern "C"
