// models.hpp -- the compiled-in models, the storage policies (placements) and the HagerZhang point type (see muse_kernels.hip).
#pragma once
#include <type_traits>

#include "step.hpp"
#include "user_model.hpp"
#include "vec.hpp"

namespace muse {

// ------------------------------------------------------------------------------------------------
// Models.  grad() returns d(-logLike)/dz_i and adds the element's share of -2 logLike (without the
// constant) to facc; the score is assembled from per-block sums of score_term().
template <int MAXB = kMaxTheta>
__device__ __forceinline__ int block_of(const BatchArgs& a, int i) {
    int k = 0;
#pragma unroll
    for (int b = 1; b < MAXB; ++b) k += (i >= a.bnd32[b]) ? 1 : 0;  // bnd32[b] = INT_MAX for b >= ntheta
    return k;
}

// (block_of_big, the big tier's block index by arithmetic: step.hpp -- host and device, checked exhaustively on the host)

template <int MAXB_>
struct FunnelModel {  // z_i ~ N(0, e^theta_k), x_i ~ N(z_i, 1)
    static constexpr int MAXB = MAXB_;
    static constexpr bool kPair = false;   // one parameter per block: the coefficients are plain doubles
    using SCoef = double; using GCoef = double;
    static constexpr bool kStencil = false;
    static constexpr int kId = MUSE_MODEL_FUNNEL;
    __device__ static __forceinline__ void sample(double sd, double n1, double n2, double& z, double& x, int) {
        z = sd * n1;
        x = z + n2;
    }
    __device__ static __forceinline__ double grad(double iv, double x, double z, double& facc, int) {
        const double r = x - z, t = iv * z;
        facc = fma(t, z, fma(r, r, facc));
        return t - r;
    }
    __device__ static __forceinline__ double score_term(double, double z, int) { return z * z; }
};
struct NoiseModel {  // z_i ~ N(0,1), x_i ~ N(z_i, e^theta)
    static constexpr int MAXB = 1;
    static constexpr bool kPair = false;   // one parameter per block: the coefficients are plain doubles
    using SCoef = double; using GCoef = double;
    static constexpr bool kStencil = false;
    static constexpr int kId = MUSE_MODEL_NOISE;
    __device__ static __forceinline__ void sample(double sd, double n1, double n2, double& z, double& x, int) {
        z = n1;
        x = n1 + sd * n2;
    }
    __device__ static __forceinline__ double grad(double iv, double x, double z, double& facc, int) {
        const double r = x - z, t = iv * r;
        facc = fma(z, z, fma(t, r, facc));
        return z - t;
    }
    __device__ static __forceinline__ double score_term(double x, double z, int) {
        const double r = x - z;
        return r * r;
    }
};
template <int MAXB_>
struct SmoothModel {  // z as funnel, x = A z + n, A = periodic (1/4, 1/2, 1/4); streaming policy only
    static constexpr int MAXB = MAXB_;
    static constexpr bool kPair = false;   // one parameter per block: the coefficients are plain doubles
    using SCoef = double; using GCoef = double;
    static constexpr bool kStencil = true;
    static constexpr int kId = MUSE_MODEL_SMOOTH;
    __device__ static __forceinline__ double score_term(double, double z, int) { return z * z; }
};

// Coefficients of an element's block for the models with TWO parameters per block (include/muse_model.h, MUSE_MODEL_PAIR): what the
// draw takes (c[0], c[1]) and what the objective takes (all four).  The tables hold a block's coefficients side by side:
// ThetaSet::sd and ::iv read as ONE array of [block][4] (pair_table), a sampling entry (SampleSd) as [block][2].
// One block (the tier of two components): plain values -- they are workgroup-uniform and live in scalar registers; the pad element
// gets zeros.  Several blocks: a POINTER to the block's record in LDS and the element's validity -- four doubles per element in
// vector registers (two elements of a pair in flight) were 500 spilled registers in the LDS-resident kernel; the model's functions
// read what they use where they use it, and the pad element's contribution is dropped behind the call.
struct PairS { double c[2]; };
struct PairGv { double c[4]; };
struct PairGp { const double* p; bool valid; };
__host__ __device__ __forceinline__ const double* pair_table(const ThetaSet& t) { return &t.sd[0]; }
static_assert(offsetof(ThetaSet, iv) == offsetof(ThetaSet, sd) + kMaxTheta * sizeof(double), "sd and iv as one [block][4] table");

#ifdef MUSE_USER_MODEL_HEADER
#ifndef MUSE_MODEL_PAIR
template <int MAXB_>
struct UserModel {  // include/muse_model.h: the three functions of the user's header behind the elementwise model concept
    static constexpr int MAXB = MAXB_;
    static constexpr bool kStencil = false;
    static constexpr int kId = MUSE_MODEL_USER;
    static constexpr bool kPair = false;
    using SCoef = double; using GCoef = double;
    __device__ static __forceinline__ void sample(double sd, double n1, double n2, double& z, double& x, int i) {
        muse_model_sample(sd, n1, n2, &z, &x, (long)i);
    }
    __device__ static __forceinline__ double grad(double iv, double x, double z, double& facc, int i) {
        return muse_model_grad(iv, x, z, &facc, (long)i);
    }
    __device__ static __forceinline__ double score_term(double x, double z, int i) { return muse_model_score_term(x, z, (long)i); }
#ifdef MUSE_MODEL_SECOND  // (include/muse_model.h: the operands of the implicit-differentiation get_H!, Solver::run_implicit)
    __device__ static __forceinline__ void second(double iv, double x, double z, double& ozz, double& ozx, double& bz, double& bx, int i) {
        muse_model_second(iv, x, z, &ozz, &ozx, &bz, &bx, (long)i);
    }
    // dx_i / dtheta_k at fixed normals: sd = exp(theta / 2)
    __device__ static __forceinline__ double dx_dtheta(double sd, double n1, double n2, int i) {
        return 0.5 * (sd * muse_model_dx_dsd(sd, n1, n2, (long)i));
    }
#endif
};
#else
// A header of the two-parameter family (include/muse_model.h, "TWO PARAMETERS PER BLOCK"): K = ntheta / 2 blocks, block k's parameters
// theta[k] and theta[K + k], four coefficients per block, two block sums per block, the score assembled by the header.
template <int MAXB_>
struct UserModel {
    static_assert(MAXB_ % 2 == 0, "two parameters per block");
    static constexpr int MAXB = MAXB_;
    static constexpr bool kStencil = false;
    static constexpr int kId = MUSE_MODEL_USER;
    static constexpr bool kPair = true;
    using SCoef = PairS;
    using GCoef = typename std::conditional<MAXB_ == 2, PairGv, PairGp>::type;
    __device__ static __forceinline__ void sample(const PairS& c, double n1, double n2, double& z, double& x, int i) {
        muse_model_sample(c.c, n1, n2, &z, &x, (long)i);
    }
    __device__ static __forceinline__ double grad(const PairGv& c, double x, double z, double& facc, int i) {
        return muse_model_grad(c.c, x, z, &facc, (long)i);
    }
    __device__ static __forceinline__ void score_terms(const PairGv& c, double x, double z, double& t0, double& t1, int i) {
        muse_model_score_terms(c.c, x, z, &t0, &t1, (long)i);
    }
    __device__ static __forceinline__ double grad(const PairGp& c, double x, double z, double& facc, int i) {
        double a2 = facc;
        const double g = muse_model_grad(c.p, x, z, &a2, (long)i);
        facc = c.valid ? a2 : facc;     // (the pad element and the phantom slots: no contribution, whatever the model's location)
        return c.valid ? g : 0.0;
    }
    __device__ static __forceinline__ void score_terms(const PairGp& c, double x, double z, double& t0, double& t1, int i) {
        muse_model_score_terms(c.p, x, z, &t0, &t1, (long)i);
        t0 = c.valid ? t0 : 0.0;
        t1 = c.valid ? t1 : 0.0;
    }
    __device__ static __forceinline__ void score(const double* c, double s0, double s1, double n, double& ga, double& gb) {
        muse_model_score(c, s0, s1, n, &ga, &gb);
    }
};
#endif
#endif

// ------------------------------------------------------------------------------------------------
// Storage policies.
template <int T_, bool CLUSTER = false, int U_ = 4, bool COH = false, bool LDS_S = false>
struct PlaceStreaming {
    static constexpr bool kCoherent = COH;  // stencil model in a cluster: see vec.hpp, kCoherent
    static constexpr bool kLdsS = LDS_S;    // the search direction in LDS (vec.hpp, LdsMirror)
    static constexpr int T = T_, EPT = 0, U = U_;  // U pairs of a thread per trip of a streaming pass
    // two waves per SIMD: 2 workgroups of 256 threads (cluster mode sizes its grid from that) or 1 of 512 per CU,
    // i.e. a budget of 256 registers per lane
    static constexpr int kWavesPerEu = 2;
    static constexpr bool kResident = false, kXgLds = false, kCluster = CLUSTER;
    using VX = BufChunk<U_, COH>;
    using VG = VX; using VZ = VX;
    using VS = typename std::conditional<LDS_S, LdsMirror<U_, T_>, VX>::type;
    using VH = VX;
};
// CLUSTER (registers only): csize workgroups share one element, thread pairs (crank*T + tid) + j*csize*T -- the
// per-GPU share of a strongly scaled map, where a launch has fewer elements than the GPU has compute units.
template <int T_, int EPT_, bool XG_LDS, bool CLUSTER = false>
struct PlaceResident {
    static_assert(!(XG_LDS && CLUSTER), "the LDS layout addresses x and g by global element index");
    static constexpr int T = T_, EPT = EPT_, U = 1;
    static constexpr int kWavesPerEu = 1;  // no lower bound beyond the launch bounds
    static constexpr bool kResident = true, kXgLds = XG_LDS, kCluster = CLUSTER, kCoherent = false, kLdsS = false;
    using VX = typename std::conditional<XG_LDS, LdsVec, RegVec<2 * EPT_>>::type;
    using VG = VX;
    using VZ = RegVec<2 * EPT_>; using VS = RegVec<2 * EPT_>;
    using VH = BufVec2;  // history vectors and zhat in HBM: 16-byte accesses
};

struct HzPoint {
    double a, v, d;  // alpha, phi(alpha), dphi(alpha)
    int id;          // evaluation sequence number (0 = the point alpha=0)
};


}  // namespace muse
