/*
 * lightloam_rccl.hpp -- the collectives of the multi-GPU mapping modes from C++, RCCL only (no torch, no MPI).
 *
 * north_star: "Host code stays C++/ROS ... RCCL all-reduce of the 6x6 / 6x1 normal equations over xGMI".  A ROS node on an
 * 8-GPU box links librccl (-lrccl, /opt/rocm/lib) next to liblightloam_hip and drives one rank per GPU -- one host thread (or
 * process) per rank -- through this header:
 *
 *   lightloam::RcclRank        one rank's communicator + the library's own HIP stream (ll_stream(ctx)): every collective is
 *                              enqueued on THAT stream, so it is ordered with the kernels of the ll_map_*_dev calls around it
 *                              and nothing synchronises until the pose is read back
 *   map_optimize_row_parallel  laserMapping.cpp:1832-2095 with the scan's stack points split over the ranks (BASELINE config 4):
 *                              associate; evaluate -> ncclAllReduce(sum, f64, 44) -> lm_begin; 4 x { propose; evaluate ->
 *                              all-reduce -> accept }.  ll_map_evaluate_dev leaves H (36), g (6), cost, rows on the device; 28 of
 *                              the 44 doubles are unique (21 + 6 + 1), the record is reduced as it is: 352 bytes, latency-bound
 *                              Returns false when the map is too small (:1822); never throws between paired collectives (see
 *                              the function's comment for the failure discipline)
 *   RcclRank::all_gather_host  the `all_gather` argument of lightloam::LaserMapping::process_tile_parallel (lightloam_host.hpp:
 *                              the map split by cube over the ranks): host buffers staged through
 *                              device memory, ncclAllGather on the library's stream
 *
 * Single-process use over all visible devices: RcclWorld (ncclCommInitAll).  Multi-process use: build the ncclComm_t yourself
 * (ncclGetUniqueId / ncclCommInitRank) and hand it to RcclRank.  tests/native/rccl_normal_equations.cpp runs both modes at
 * whatever world size the box offers (1 on the one-GPU pool) and compares with the one-rank calls.
 */
#pragma once
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <stdexcept>
#include <string>
#include <vector>

#include "lightloam_hip.h"

namespace lightloam {

struct RcclError : std::runtime_error {
    explicit RcclError(const std::string &what) : std::runtime_error(what) {}
};
#define LL_RCCL_CHECK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) throw ::lightloam::RcclError(std::string(#call) + ": " + ncclGetErrorString(r_)); } while (0)
#define LL_RCCL_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) throw ::lightloam::RcclError(std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

/* one communicator per visible device, one process (ncclCommInitAll) */
class RcclWorld {
public:
    explicit RcclWorld(int world = 0) {
        int n = 0;
        LL_RCCL_HIP(hipGetDeviceCount(&n));
        if (world <= 0) world = n;
        if (world > n) throw RcclError("more ranks than visible devices");
        comms_.resize((size_t)world);
        std::vector<int> devs((size_t)world);
        for (int r = 0; r < world; ++r) devs[(size_t)r] = r;
        LL_RCCL_CHECK(ncclCommInitAll(comms_.data(), world, devs.data()));
    }
    ~RcclWorld() { for (ncclComm_t c : comms_) if (c) (void)ncclCommDestroy(c); }
    RcclWorld(const RcclWorld &) = delete;
    RcclWorld &operator=(const RcclWorld &) = delete;
    int size() const { return (int)comms_.size(); }
    ncclComm_t comm(int rank) const { return comms_[(size_t)rank]; }
    /* a communicator that RcclRank::abort() has already torn down must not be destroyed again: a rank built with this world as
     * its owner calls forget() from abort() itself */
    void release(int rank) { comms_[(size_t)rank] = nullptr; }
    void forget(ncclComm_t c) noexcept { for (ncclComm_t &x : comms_) if (x == c) x = nullptr; }
private:
    std::vector<ncclComm_t> comms_;
};

/* a rank: its communicator, the context whose stream carries the collectives, device staging for host-side gathers */
class RcclRank {
public:
    /* owner: the RcclWorld the communicator came from (nullptr: the caller built it and destroys it); abort() tells it that the
     * communicator is gone, so that ~RcclWorld does not destroy it a second time whichever way the stack unwinds */
    RcclRank(ncclComm_t comm, ll_ctx *ctx, int device, RcclWorld *owner = nullptr) : comm_(comm), stream_((hipStream_t)ll_stream(ctx)), device_(device), owner_(owner) {
        LL_RCCL_CHECK(ncclCommCount(comm_, &world_));
        LL_RCCL_CHECK(ncclCommUserRank(comm_, &rank_));
        LL_RCCL_HIP(hipSetDevice(device_));
        LL_RCCL_HIP(hipMalloc(&neq_, 44 * sizeof(double)));
    }
    ~RcclRank() { (void)hipSetDevice(device_); if (neq_) (void)hipFree(neq_); if (send_) (void)hipFree(send_); if (recv_) (void)hipFree(recv_); }
    RcclRank(const RcclRank &) = delete;
    RcclRank &operator=(const RcclRank &) = delete;
    int rank() const { return rank_; }
    int world() const { return world_; }
    hipStream_t stream() const { return stream_; }
    double *neq_dev() const { return (double *)neq_; }
    /* the sum over the ranks of what ll_map_evaluate_dev left in neq_dev(), in place, on the library's stream */
    void all_reduce_neq() { LL_RCCL_CHECK(ncclAllReduce(neq_, neq_, 44, ncclDouble, ncclSum, comm_, stream_)); ++n_allreduce; }
    bool try_all_reduce_neq() noexcept { if (ncclAllReduce(neq_, neq_, 44, ncclDouble, ncclSum, comm_, stream_) != ncclSuccess) return false; ++n_allreduce; return true; }
    /* give up on the communicator: the peers' pending collectives return with an error instead of waiting for this rank */
    void abort() noexcept {
        if (comm_ && !aborted_) {
            (void)ncclCommAbort(comm_); aborted_ = true;
            if (owner_) owner_->forget(comm_);
        }
    }
    bool aborted() const { return aborted_; }
    /* `bytes` from every rank into recv, rank-major; host buffers (the shape LaserMapping::process_tile_parallel asks for) */
    void all_gather_host(const void *send, void *recv, size_t bytes) {
        LL_RCCL_HIP(hipSetDevice(device_));
        if (bytes > cap_) {
            if (send_) (void)hipFree(send_);
            if (recv_) (void)hipFree(recv_);
            cap_ = bytes * 2;
            LL_RCCL_HIP(hipMalloc(&send_, cap_));
            LL_RCCL_HIP(hipMalloc(&recv_, cap_ * (size_t)world_));
        }
        LL_RCCL_HIP(hipMemcpyAsync(send_, send, bytes, hipMemcpyHostToDevice, stream_));
        LL_RCCL_CHECK(ncclAllGather(send_, recv_, bytes, ncclChar, comm_, stream_));
        LL_RCCL_HIP(hipMemcpyAsync(recv, recv_, bytes * (size_t)world_, hipMemcpyDeviceToHost, stream_));
        LL_RCCL_HIP(hipStreamSynchronize(stream_));
        ++n_allgather;
    }
    long n_allreduce = 0, n_allgather = 0;
private:
    ncclComm_t comm_;
    hipStream_t stream_;
    int device_, world_ = 1, rank_ = 0;
    void *neq_ = nullptr, *send_ = nullptr, *recv_ = nullptr;
    size_t cap_ = 0;
    bool aborted_ = false;
    RcclWorld *owner_ = nullptr;
};

/* laserMapping.cpp:1822-2095 row-parallel: `m` holds the whole map and THIS rank's slice of the stack clouds (ll_map_set_scan).
 * pose_w7: the guess on entry (the same on every rank), the optimised q_w_curr / t_w_curr on return (bit-identical on every rank:
 * all ranks step the same Levenberg-Marquardt state with the same sums).  One host upload, one read-back; the rest is enqueued.
 * Returns false -- pose_w7 untouched -- when the map is too small to optimise against (:1822: <= 10 corner or <= 50 surf points; the
 * reference then keeps the odometry guess, :2096-2100): every rank holds the same map, so every rank takes that exit before the
 * first collective.
 *
 * Failure discipline (nothing here may leave the sequence between paired collectives: a rank that stopped calling ncclAllReduce
 * would leave its peers blocked in theirs, or in the stream sync of ll_map_get_pose):
 *   - a failed ll_map_* call on THIS rank is remembered, the rank skips its remaining ll_map_* steps but still issues every
 *     all-reduce of the sequence, with a record of NaNs (0xFF bytes): the sum is NaN on every rank, every rank's LM state and
 *     pose become NaN, and ll_map_get_pose reports LL_ERR_STATE everywhere -- all ranks leave together, each with an error;
 *   - a failed collective or HIP call (the stream or communicator itself is broken) aborts the communicator (ncclCommAbort),
 *     which releases the peers from their pending collectives with an error of their own.
 * Only after the sequence is complete does the rank throw. */
inline bool map_optimize_row_parallel(ll_map *m, RcclRank &rk, double pose_w7[7], int n_outer = 2, const ll_lm_options *opt = nullptr)
{
    ll_lm_options o;
    if (opt) o = *opt; else ll_lm_default_options(&o);
    int n_corner = 0, n_surf = 0;
    if (ll_map_get_map_sizes(m, &n_corner, &n_surf) != LL_OK) throw RcclError("lightloam: ll_map_get_map_sizes failed");   /* host state, before any collective */
    if (!(n_corner > 10 && n_surf > 50)) return false;                 /* :1822 */
    std::string local_err;                                             /* first failure of an ll_map_* call on this rank */
    auto step = [&](auto &&call) { if (local_err.empty() && call() != LL_OK) local_err = std::string("lightloam: ") + ll_map_last_error(m); };
    auto reduce = [&]() {                                              /* never skipped */
        if (!local_err.empty()) (void)hipMemsetAsync(rk.neq_dev(), 0xFF, 44 * sizeof(double), rk.stream());   /* NaN record */
        if (!rk.try_all_reduce_neq()) { rk.abort(); throw RcclError("rccl: all-reduce failed, communicator aborted" + (local_err.empty() ? std::string() : " after " + local_err)); }
    };
    step([&] { return ll_map_set_pose(m, pose_w7); });
    for (int it = 0; it < n_outer; ++it) {
        step([&] { return ll_map_associate(m, nullptr); });            /* at the device pose */
        step([&] { return ll_map_evaluate_dev(m, rk.neq_dev()); }); reduce();
        step([&] { return ll_map_lm_begin_dev(m, rk.neq_dev(), &o); });
        for (int k = 0; k < o.max_num_iterations; ++k) {
            step([&] { return ll_map_lm_propose_dev(m, &o); });
            step([&] { return ll_map_evaluate_dev(m, rk.neq_dev()); }); reduce();
            step([&] { return ll_map_lm_accept_dev(m, rk.neq_dev(), &o); });
        }
    }
    if (!local_err.empty()) throw RcclError(local_err);                /* the peers see NaN sums and fail in ll_map_get_pose */
    double out[7];
    if (ll_map_get_pose(m, out) != LL_OK)                              /* the one synchronising read-back; LL_ERR_STATE = a peer's NaN record */
        throw RcclError(std::string("lightloam: ") + ll_map_last_error(m));
    for (int i = 0; i < 7; ++i) pose_w7[i] = out[i];
    return true;
}

}  // namespace lightloam
