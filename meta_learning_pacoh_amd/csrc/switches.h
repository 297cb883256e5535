// Every PACOH_* environment switch the library honours (test / A-B switches), read ONCE when the library is
// loaded -- no getenv on a launch path.  A host that changes the environment afterwards (the path-forcing tests,
// tools/mlp_time.py) calls pacoh_reload_env(), which reads it again.  Defined in misc.hip.
#pragma once

namespace pacoh {

struct Switches {
    // large-context (HBM-resident) path
    bool chol_ll, trtri_ll, retry_fused, gemm_tile, trtri_blocked, chol_blocked, grad_mfma, grad_mfma_f32, gram_mfma, dense_pad;
    bool mfma;                 // PACOH_DISABLE_MFMA=1 -> false: the next GP / Cholesky / MLP implementation (fallback coverage)
    // register-resident GP kernels
    bool gp_reg, gp_reg_predict;
    bool gp8;                  // PACOH_GP8=0: the task-fused / persistent kernels run contexts of <= 8 points on the block body too (A/B, tests)
    int gp_reg_max_n;
    // per-particle MLP
    bool fused_mlp;
    int mlp_path;              // 0 fused, 1 mfma, 3 layers: the first implementation the dispatcher may pick
    int mlp_stash, fused_bwd_pb, fused_fwd_pb, fused_fwd_tpw;      // -1 / 0: the dispatcher's own choice
    // A/B: extra dynamic LDS per workgroup of the register-resident GP kernel / the fused MLP kernels (bytes; -1: the launchers' own
    // choice) -- caps the workgroups a CU takes, i.e. spreads an under-filled grid over all CUs (spread_pad, common.h)
    int lds_pad_gp, lds_pad_mlp;
    int mt_nt;                 // PACOH_MT_NT=512 | 1024: threads per workgroup of the task-fused kernels (A/B; anything else: by the number of weight tiles)
};

extern Switches g_sw;
void read_switches(Switches& s);

}  // namespace pacoh
