// Host-side AddressSanitizer shim (sanitizers on the CPU build only).  Linked against a build of the
// library whose HOST code is compiled with -fsanitize=address (device code left out: --cuda-host-only), it walks the part of the
// C ABI that runs entirely on the host -- argument validation, limit checks, the *_workspace_bytes / *_stash_bytes / *_max_n
// queries, the dispatchers that read the caller's `hidden[]` array -- with exactly-sized heap buffers, so that a read past the end
// of a caller array or a use of freed scratch inside those paths aborts the run.  Nothing here launches a kernel or needs a GPU.
// Mirrors tests/test_abi.py::test_argument_validation_returns_error_codes_without_launching.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pacoh_gp.h"

static int failures = 0;
#define EXPECT(cond) do { if (!(cond)) { std::fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #cond); ++failures; } } while (0)

static int32_t* heap_hidden(std::initializer_list<int> w) {            // exactly n_hidden elements: ASan guards the byte after
    int32_t* p = static_cast<int32_t*>(std::malloc(sizeof(int32_t) * (w.size() ? w.size() : 1)));
    int i = 0;
    for (int v : w) p[i++] = v;
    return p;
}

int main() {
    void* fake = reinterpret_cast<void*>(4096);                          // non-NULL, never dereferenced on these paths
    if (std::getenv("PACOH_ASAN_SELFCHECK")) {
        // proof that the instrumentation is live: the library is told about one more layer than the caller's array holds, and its
        // host-side dispatcher reads hidden[2] of a 2-element heap array -> AddressSanitizer must abort the process
        int32_t* h = heap_hidden({32, 32});
        std::printf("%zu\n", pacoh_mlp_bwd_workspace_bytes(60, 3, 33, 2, h, 3, 2, PACOH_F32));
        return 0;
    }
    EXPECT(pacoh_abi_version() == 14);
    pacoh_reload_env();                                   // (the switches are read at load time and here, never on a launch path)
    EXPECT(pacoh_gp_small_max_n(PACOH_F32, 0) >= 128 && pacoh_gp_small_max_n(PACOH_F64, 1) >= 64 && pacoh_gp_small_max_n(7, 0) == PACOH_EDTYPE);
    EXPECT(pacoh_svgd_workspace_bytes(20, 2534, PACOH_F32) == (2 * 400 + 20 + 8) * 4);
    EXPECT(pacoh_gp_predict_workspace_bytes(4, 64, 50, PACOH_F32, 1) == 4u * 50 * 64 * 4 && pacoh_gp_predict_workspace_bytes(4, 64, 50, PACOH_F32, 0) == 0);
    EXPECT(pacoh_gp_lml_dense_workspace_bytes(4, 512, 8, PACOH_F64, 1) > 4u * 512 * 512 * 8);
    EXPECT(pacoh_gp_predict_dense_workspace_bytes(4, 512, 100, PACOH_F64) > 0);
    EXPECT(pacoh_vi_update_dev_workspace_bytes(2534, PACOH_F32) > 0 && pacoh_svgd_update_dev_workspace_bytes(20, 2534, PACOH_F32) > 0);
    EXPECT(pacoh_svgd_imq_workspace_bytes(20, 2534, PACOH_F64) > 0);

    // every dispatcher of the per-particle MLP reads hidden[0 .. n_hidden): exactly-sized heap arrays, all four paths
    struct Case { std::initializer_list<int> w; int d_in, d_out; };
    const Case cases[] = {{{32, 32}, 4, 2}, {{32, 32, 32, 32}, 1, 1}, {{32, 32}, 9, 5}, {{64, 64, 64}, 2, 2}, {{128, 128, 128, 128}, 1, 2},
                          {{7}, 3, 1}, {{}, 2, 5}, {{20, 32, 11, 5, 9, 300}, 3, 2}};
    for (const Case& c : cases) {
        int32_t* h = heap_hidden(c.w);
        const int nh = static_cast<int>(c.w.size());
        for (int dt = 0; dt < 2; ++dt) {
            const size_t fw = pacoh_mlp_fwd_workspace_bytes(60, 3, 33, c.d_in, h, nh, c.d_out, dt);
            const size_t bw = pacoh_mlp_bwd_workspace_bytes(60, 3, 33, c.d_in, h, nh, c.d_out, dt);
            EXPECT(bw > 0);
            EXPECT(pacoh_mlp2_fwd_workspace_bytes(60, 3, 33, c.d_in, h, nh, 1, c.d_out, dt) >= (c.d_out >= 1 ? fw : 0));
            EXPECT(pacoh_mlp2_bwd_workspace_bytes(60, 3, 33, c.d_in, h, nh, 1, c.d_out, dt) > 0);
            (void)pacoh_mlp2_stash_bytes(60, 3, 33, c.d_in, h, nh, 1, c.d_out, dt);
            // argument errors are reported before anything is launched or dereferenced
            EXPECT(pacoh_mlp_fwd(nullptr, 1, fake, 10, 3, c.d_in, h, nh, c.d_out, fake, nullptr, 60, 33, dt, nullptr) == PACOH_EINVAL);
            EXPECT(pacoh_mlp_bwd(fake, 1, fake, 10, 3, c.d_in, h, nh, c.d_out, nullptr, fake, 10, 0, fake, 60, 33, dt, nullptr) == PACOH_EINVAL);
            EXPECT(pacoh_mlp_fwd(fake, 1, fake, 10, 3, c.d_in, h, nh, c.d_out, fake, nullptr, 61, 33, dt, nullptr) == PACOH_EINVAL);   // B % P != 0
        }
        std::free(h);
    }
    {
        int32_t* h = heap_hidden({32, 32, 32, 32});
        EXPECT(pacoh_mlp2_stash_bytes(20, 10, 20, 1, h, 4, 1, 2, PACOH_F32) == 3u * 2 * 10 * 4 * 2048);
        EXPECT(pacoh_mlp2_stash_bytes(20, 10, 20, 1, h, 4, 1, 2, PACOH_F64) > 0);     // (round 6: the layer-by-layer path keeps packed weights + activations)
        std::free(h);
        h = heap_hidden({1 << 20});
        EXPECT(pacoh_mlp_fwd(fake, 1, fake, 10, 1, 2, h, 1, 1, fake, nullptr, 1, 4, PACOH_F32, nullptr) == PACOH_ELIMIT);
        std::free(h);
        h = heap_hidden({128, 128, 128, 128});
        EXPECT(pacoh_mlp_fwd_workspace_bytes(10, 1, 5, 1, h, 4, 2, PACOH_F32) > 0);
        EXPECT(pacoh_mlp_fwd(fake, 1, fake, 10, 1, 1, h, 4, 2, fake, nullptr, 10, 5, PACOH_F32, nullptr) == PACOH_EINVAL);   // needs a workspace
        std::free(h);
    }
    {   // the persistent PACOH-MAP kernel's host-side plan reads mean_hidden[] / kernel_hidden[] of exactly n entries (round 5)
        int32_t* hm = heap_hidden({32, 32});
        int32_t* hk = heap_hidden({16});
        EXPECT(pacoh_map_persist_supported(5, 1, 5, PACOH_MEAN_VECTOR, hm, 2, 1, hk, 1, 2, PACOH_F32) == 1);
        EXPECT(pacoh_map_persist_supported(5, 1, 17, PACOH_MEAN_VECTOR, hm, 2, 1, hk, 1, 2, PACOH_F32) == 0);     // more tasks than waves
        EXPECT(pacoh_map_persist_supported(5, 1, 5, PACOH_MEAN_VECTOR, hm, 2, 0, nullptr, 0, 1, PACOH_F64) == 0);
        int32_t seg[4] = {0, 0, 0, 0};
        EXPECT(pacoh_map_persist(fake, fake, fake, 8, fake, fake, nullptr, 33, 1, (const int64_t*)fake, 5, fake, PACOH_SC_COUNT, 1, PACOH_MEAN_ZERO, -1,
                                 nullptr, 0, 0, -1, nullptr, 0, 1, 0, 1, 2, 1e-3, seg, seg, 1, 0.9, 0.999, nullptr, nullptr, nullptr, PACOH_F32,
                                 nullptr) == PACOH_ELIMIT);                                                                          // n > 32
        EXPECT(pacoh_map_persist(nullptr, fake, fake, 8, fake, fake, nullptr, 5, 1, (const int64_t*)fake, 5, fake, PACOH_SC_COUNT, 1, PACOH_MEAN_ZERO, -1,
                                 nullptr, 0, 0, -1, nullptr, 0, 1, 0, 1, 2, 1e-3, seg, seg, 1, 0.9, 0.999, nullptr, nullptr, nullptr, PACOH_F32,
                                 nullptr) == PACOH_EINVAL);
        // a parameter layout that leaves the row: the persistent kernel reads AND writes theta / both moments at these offsets
        {
            int32_t lo[4] = {0, 0, 0, 0}, hi[4] = {40, 0, 0, 0};
            const int Dm = 32 * 2 + 32 * 33 + 33;                    // NN(32, 32) mean on d = 1: 1153 columns from off_mean
            EXPECT(pacoh_map_persist(fake, fake, fake, Dm + 3, fake, fake, nullptr, 5, 1, (const int64_t*)fake, 5, fake, PACOH_SC_COUNT, 1, PACOH_MEAN_VECTOR, 8,
                                     hm, 2, 0, -1, nullptr, 0, 1, 0, -1, 2, 1e-3, lo, hi, 1, 0.9, 0.999, nullptr, nullptr, nullptr, PACOH_F32,
                                     nullptr) == PACOH_EINVAL);                                                                      // network block ends at 8 + 1153 > D
            EXPECT(pacoh_map_persist(fake, fake, fake, 40, fake, fake, nullptr, 5, 1, (const int64_t*)fake, 5, fake, PACOH_SC_COUNT, 1, PACOH_MEAN_ZERO, -1,
                                     nullptr, 0, 0, -1, nullptr, 0, 1, 40, -1, 2, 1e-3, lo, hi, 1, 0.9, 0.999, nullptr, nullptr, nullptr, PACOH_F32,
                                     nullptr) == PACOH_EINVAL);                                                                      // off_ls == D
            hi[0] = 41;
            EXPECT(pacoh_map_persist(fake, fake, fake, 40, fake, fake, nullptr, 5, 1, (const int64_t*)fake, 5, fake, PACOH_SC_COUNT, 1, PACOH_MEAN_ZERO, -1,
                                     nullptr, 0, 0, -1, nullptr, 0, 1, 0, -1, 2, 1e-3, lo, hi, 1, 0.9, 0.999, nullptr, nullptr, nullptr, PACOH_F32,
                                     nullptr) == PACOH_EINVAL);                                                                      // trained range beyond D
        }
        // the task-fused kernels are RBF-only: another kernel family must not silently train with the RBF Gram
        const int f_cos = 1 | (PACOH_KERNEL_COSINE << PACOH_KERNEL_SHIFT);
        EXPECT(pacoh_map_task_workspace_bytes(2000, 32, 1, 256, PACOH_MEAN_VECTOR, hm, 2, 0, nullptr, 0, 1, 0, PACOH_F32) > 0);
        EXPECT(pacoh_map_task_workspace_bytes(2000, 32, 1, 256, PACOH_MEAN_VECTOR, hm, 2, 0, nullptr, 0, f_cos, 1, PACOH_F32) == 0);
        EXPECT(pacoh_map_task_setup(fake, 2000, 32, 1, 256, PACOH_MEAN_VECTOR, 0, hm, 2, 0, -1, nullptr, 0, f_cos, fake, 1 << 20, PACOH_F32, nullptr) == PACOH_ELIMIT);
        EXPECT(pacoh_map_task_step(fake, 2000, fake, fake, nullptr, 32, 1, 256, PACOH_MEAN_VECTOR, 0, hm, 2, 0, -1, nullptr, 0, f_cos, fake, nullptr, fake,
                                   1990, -1, 1991, fake, 2000, nullptr, 1.0, nullptr, fake, 1 << 20, nullptr, PACOH_F32, nullptr) == PACOH_ELIMIT);
        // the same kernel with P parameter rows (round 6)
        EXPECT(pacoh_svgd_task_workspace_bytes(2534, 10, 20, 1, 2, PACOH_MEAN_VECTOR, hm, 2, 1, hm, 2, 2, 0, PACOH_F32) > 0);     // (no device: the plan alone decides)
        EXPECT(pacoh_svgd_task_workspace_bytes(2534, 10, 33, 1, 2, PACOH_MEAN_VECTOR, hm, 2, 1, hm, 2, 2, 0, PACOH_F32) == 0);
        EXPECT(pacoh_svgd_task_workspace_bytes(2534, 10, 20, 1, 2, PACOH_MEAN_VECTOR, hm, 2, 1, hm, 2, 2 | (PACOH_KERNEL_COSINE << PACOH_KERNEL_SHIFT), 1, PACOH_F32) == 0);
        EXPECT(pacoh_svgd_task_step(nullptr, 2534, 10, fake, fake, nullptr, 20, 1, 2, PACOH_MEAN_VECTOR, 0, hm, 2, 1, 1200, hm, 2, 2, fake, nullptr, fake,
                                    2530, -1, 2533, fake, 2534, fake, 1.0, nullptr, fake, 1 << 20, nullptr, nullptr, 0, nullptr, 0, PACOH_F32,
                                    nullptr) == PACOH_EINVAL);
        EXPECT(pacoh_svgd_task_step(fake, 2534, 10, fake, fake, nullptr, 20, 1, 2, PACOH_MEAN_VECTOR, 0, hm, 2, 1, 1200, hm, 2, 2, fake, nullptr, fake,
                                    2530, -1, 2533, fake, 2534, fake, 1.0, nullptr, fake, 1 << 20, fake, nullptr, 2534, nullptr, 0, PACOH_F32,
                                    nullptr) == PACOH_EINVAL);                                                                       // svgd_X without its workspace
        EXPECT(pacoh_svgd_task_step(fake, 2534, 10, fake, fake, nullptr, 20, 1, 2, PACOH_MEAN_VECTOR, 0, hm, 2, 1, 1200, hm, 2, 2, fake, nullptr, fake,
                                    2530, -1, 2533, fake, 2534, fake, 1.0, nullptr, fake, 1 << 20, nullptr, nullptr, 0, nullptr, 0, PACOH_F64,
                                    nullptr) == PACOH_ELIMIT);
        // hyper-parameter / network offsets beyond the parameter row: an error, not a device write out of bounds
        EXPECT(pacoh_svgd_task_step(fake, 2534, 10, fake, fake, nullptr, 20, 1, 2, PACOH_MEAN_VECTOR, 0, hm, 2, 1, 1200, hm, 2, 2, fake, nullptr, fake,
                                    2533, -1, 2533, fake, 2534, fake, 1.0, nullptr, fake, 1 << 20, nullptr, nullptr, 0, nullptr, 0, PACOH_F32,
                                    nullptr) == PACOH_EINVAL);                                                                       // off_ls + f > row
        EXPECT(pacoh_svgd_task_step(fake, 2534, 10, fake, fake, nullptr, 20, 1, 2, PACOH_MEAN_VECTOR, 0, hm, 2, 1, 1200, hm, 2, 2, fake, nullptr, fake,
                                    2530, -1, 2533, fake, 2000, fake, 1.0, nullptr, fake, 1 << 20, nullptr, nullptr, 0, nullptr, 0, PACOH_F32,
                                    nullptr) == PACOH_EINVAL);                                                                       // d_theta rows shorter than the offsets
        EXPECT(pacoh_svgd_task_step(fake, 2534, 10, fake, fake, nullptr, 20, 1, 2, PACOH_MEAN_VECTOR, -4, hm, 2, 1, 1200, hm, 2, 2, fake, nullptr, fake,
                                    2530, -1, 2533, fake, 2534, fake, 1.0, nullptr, fake, 1 << 20, nullptr, nullptr, 0, nullptr, 0, PACOH_F32,
                                    nullptr) == PACOH_EINVAL);                                                                       // negative network offset
        EXPECT(pacoh_map_task_step(fake, 2000, fake, fake, nullptr, 32, 1, 256, PACOH_MEAN_VECTOR, 0, hm, 2, 0, -1, nullptr, 0, 1, fake, nullptr, fake,
                                   1990, -1, 2000, fake, 2000, nullptr, 1.0, nullptr, fake, 1 << 20, nullptr, PACOH_F32, nullptr) == PACOH_EINVAL);   // off_noise == row
        std::free(hm); std::free(hk);
    }
    EXPECT(pacoh_gp_lml_fwd(nullptr, 1, nullptr, 0, nullptr, 1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 4, 1, 16, 2,
                            PACOH_F32, nullptr) == PACOH_EINVAL);
    EXPECT(pacoh_gp_lml_fwd(nullptr, 1, nullptr, 0, nullptr, 1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 4, 1, 16, 2,
                            5, nullptr) == PACOH_EDTYPE);
    EXPECT(pacoh_gram_rbf_ard(nullptr, 1, nullptr, 1, nullptr, nullptr, nullptr, 0, nullptr, 1, 1, 8, 8, 2, PACOH_F32, nullptr) == PACOH_EINVAL);
    EXPECT(pacoh_gram_rbf_ard(fake, 1, fake, 1, fake, nullptr, nullptr, 0, fake, 1, 1, 8, 8, 17, PACOH_F32, nullptr) == PACOH_ELIMIT);
    EXPECT(pacoh_gp_lml_fwdbwd(fake, 1, nullptr, 0, fake, 1, fake, nullptr, fake, nullptr, nullptr, fake, nullptr, nullptr, fake, nullptr, fake,
                               nullptr, 2, 1, 4096, 2, PACOH_F32, nullptr) == PACOH_ELIMIT);
    EXPECT(pacoh_svgd_phi(fake, fake, 0.0, 0, fake, nullptr, fake, 1025, 10, PACOH_F32, nullptr) == PACOH_ELIMIT);
    // the pipelined SVGD step (round 3): argument validation happens before any launch
    EXPECT(pacoh_svgd_dist_advance(nullptr, fake, 5, 10, nullptr, PACOH_F32, nullptr) == PACOH_EINVAL);
    EXPECT(pacoh_svgd_dist_advance(fake, fake, 1025, 10, nullptr, PACOH_F32, nullptr) == PACOH_ELIMIT);
    EXPECT(pacoh_svgd_update_next(fake, fake, nullptr, nullptr, 0.1, -1.0, 1, 0.9, 0.999, fake, fake, nullptr, fake, 5, 10,
                                  nullptr, fake, PACOH_SC_COUNT, nullptr, 0, fake, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0,
                                  0, 2, -1, 2, 0.0, nullptr, nullptr, nullptr, 0, PACOH_F32, nullptr) == PACOH_EINVAL);       // no counter
    EXPECT(pacoh_svgd_update_next(fake, fake, nullptr, nullptr, 0.1, -1.0, 1, 0.9, 0.999, fake, fake, nullptr, fake, 5, 10,
                                  (const int64_t*)fake, fake, PACOH_SC_COUNT, nullptr, 0, fake, nullptr, nullptr, nullptr, nullptr, nullptr,
                                  nullptr, 0, 0, 8, 3, -1, 2, 0.0, fake, nullptr, fake, 0, PACOH_F32, nullptr) == PACOH_EINVAL);   // ls beyond D
    EXPECT(pacoh_hyper_bwd(fake, 10, 3, 2, 0, 2, -1, 2, -1, fake, nullptr, fake, nullptr, fake, 10, nullptr, nullptr, 1.0, nullptr, nullptr,
                           fake, 65, 10, nullptr, PACOH_F32, nullptr) == PACOH_ELIMIT);                     // bandwidth block: register sort, P <= 64
    {
        int32_t* h = (int32_t*)std::malloc(2 * sizeof(int32_t));
        h[0] = h[1] = 32;
        EXPECT(pacoh_mlp2_fwd_svgd(fake, 1, fake, 10, 3, 2, h, 2, 0, 1, fake, 0, 2, fake, nullptr, nullptr, 60, 33, nullptr, fake, 3, 10, nullptr,
                                   PACOH_F32, nullptr) == PACOH_EINVAL);                                                       // no particles
        std::free(h);
    }
    EXPECT(pacoh_step_begin_vi(nullptr, 0, fake, PACOH_SC_COUNT, fake, 59, (int64_t*)fake, (int32_t*)fake, fake, fake, nullptr, nullptr, nullptr,
                               nullptr, nullptr, nullptr, 0, 0, fake, 6, 10, fake, fake, 0, 2, -1, 2, 0.0, nullptr, nullptr, nullptr, 0,
                               PACOH_F32, nullptr) == PACOH_EINVAL);                              // noise row is not S x D values
    {   // the AdamW step folded into the gradient epilogue: one parameter row only, and the likelihood sums must be requested
        pacoh_adam_inline opt = {fake, fake, fake, fake, 0.9, 0.999, 1, {0, 0, 0, 0}, {5, 0, 0, 0}, nullptr, nullptr, nullptr};
        EXPECT(pacoh_hyper_bwd(fake, 10, 3, 2, 0, 2, -1, 2, -1, fake, nullptr, fake, nullptr, fake, 10, fake, fake, 1.0, nullptr, nullptr,
                               nullptr, 0, 0, &opt, PACOH_F32, nullptr) == PACOH_EINVAL);           // P = 3
        EXPECT(pacoh_hyper_bwd(fake, 10, 1, 2, 0, 2, -1, 2, -1, fake, nullptr, fake, nullptr, fake, 10, nullptr, nullptr, 1.0, nullptr, nullptr,
                               nullptr, 0, 0, &opt, PACOH_F32, nullptr) == PACOH_EINVAL);           // no likelihood block to count the step
    }
    EXPECT(pacoh_adam_step(nullptr, nullptr, nullptr, nullptr, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 10, PACOH_F32, nullptr) == PACOH_EINVAL);
    EXPECT(pacoh_allreduce_sum(nullptr, 4, PACOH_F32, fake, nullptr) == PACOH_EINVAL);
    EXPECT(pacoh_allreduce_sum(fake, 4, 7, fake, nullptr) == PACOH_EDTYPE);
    EXPECT(pacoh_comm_init(nullptr, 0, 1, nullptr) == PACOH_EINVAL);
    if (failures) { std::fprintf(stderr, "%d check(s) failed\n", failures); return 1; }
    std::puts("ASAN SHIM OK");
    return 0;
}
