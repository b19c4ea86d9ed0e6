// Host-side exerciser of libmpl_hip.so for the ASan + UBSan job (tests/test_sanitize_cpu.py; SURVEY.md section 5 "sanitizers").
// The library is built HOST-ONLY for it (hipcc --cuda-host-only -fsanitize=address,undefined: no device code objects, GPU
// sanitizers are not available on this pool) and this program walks what the host side of csrc/api.hip does WITHOUT a GPU:
// argument validation, workspace carving arithmetic, the schedule arrays, struct marshalling, the error strings and the
// per-device state (mutex / event chain, error word, profile brackets).  Every entry point that needs a device must come
// back with an MPL_E_* code -- never a crash, a leak, an out-of-bounds access or undefined behaviour.
// Test infrastructure: nothing under openmpl_amd/ uses it.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mpl_hip.h"

static int failures = 0;
#define EXPECT(cond)                                                        \
    do {                                                                    \
        if (!(cond)) {                                                      \
            std::fprintf(stderr, "EXPECT failed at line %d: %s\n", __LINE__, #cond); \
            ++failures;                                                     \
        }                                                                   \
    } while (0)

int main() {
    EXPECT(mpl_hip_abi_version() == MPL_HIP_ABI_VERSION);
    for (int c = -9; c <= 1; ++c) {
        const char* s = mpl_hip_error_string(c);
        EXPECT(s != nullptr && std::strlen(s) > 0);
    }

    // ---- configuration queries: every flag combination, degenerate and huge batches
    for (unsigned f = 0; f < (1u << 13); f += 37) {
        mpl_config cfg{17, 32, 12, 8, 4, (f & 1) ? 3 : 2, f, 0};
        const int w = mpl_fpt_width(&cfg);
        EXPECT(w == 32 || w == 544 || w == 1088 || w < 0);
        for (int B : {0, 1, 3, 1024, 8192, 1 << 20}) (void)mpl_forward_workspace_bytes(&cfg, B);
    }
    {
        mpl_config bad{0, 0, 0, 0, 0, 0, 0, 0};
        (void)mpl_fpt_width(&bad);
        (void)mpl_forward_workspace_bytes(&bad, 16);
        (void)mpl_fpt_width(nullptr);
        (void)mpl_forward_workspace_bytes(nullptr, 16);
        mpl_config neg{17, 32, -1, 8, 4000, 2, 0, 0};
        (void)mpl_forward_workspace_bytes(&neg, -5);
    }
    for (int n_seq : {0, 1, 256, 1024, 1 << 22})
        for (int n_tok : {0, 1, 2, 4, 31, 32, 33, 527})
            for (int dim : {0, 32, 136, 544, 1088, 4352, 7}) (void)mpl_block_stack_workspace_bytes(n_seq, n_tok, dim);
    for (int N : {0, 51, 96, 136, 544, 1088, 1632, 2176, 3264})
        for (int K : {0, 32, 64, 100, 544, 1088, 2176}) {
            (void)mpl_pack_h2_bytes(N, K);
            (void)mpl_pack_bf16_bytes(N, K);
            (void)mpl_ln_linear_h2_workspace_bytes(N, K);
        }
    EXPECT(mpl_pack_h2_bytes(1632, 544) == (size_t)12 * 17 * 18 * 1024 + (5 * 1632 + 8) * 4);
    EXPECT(mpl_pack_bf16_bytes(544, 100) == 0);
    EXPECT(mpl_spt_pack_bytes() >= 34 * 1024);
    EXPECT(mpl_pose_metrics_size(17) > 0);
    (void)mpl_pose_metrics_size(0);
    (void)mpl_pose_metrics_size(-3);
    (void)mpl_pack_h2_out_scale(nullptr, 1632, 544);

    // ---- the launch rule query: invalid arguments are refused before any device query; valid ones need a device
    EXPECT(mpl_block_stack_form(0, 2, 544, 8, 13, 2, 0) < 0);
    EXPECT(mpl_block_stack_form(256, 0, 544, 8, 13, 2, 0) < 0);
    EXPECT(mpl_block_stack_form(256, 2, 0, 8, 13, 2, 0) < 0);
    EXPECT(mpl_block_stack_form(256, 2, 544, 0, 13, 2, 0) < 0);
    EXPECT(mpl_block_stack_form(256, 2, 544, 8, 0, 2, 0) < 0);
    EXPECT(mpl_block_stack_form(256, 2, 544, 8, 13, 7, 0) < 0);
    (void)mpl_block_stack_form(256, 2, 544, 8, 13, 2, 0);
    (void)mpl_block_stack_form(1, 2, 544, 8, 13, 2, MPL_F_NO_SMALL_STACK);
    (void)mpl_block_stack_form(1024, 4, 1088, 8, MPL_MAX_APPS + 5, 1, 0);
    EXPECT(mpl_block_stack_form_ex(1, 2, 544, 8, 13, 14, 1, 2, 0) < 0);       // more blocks than applications
    EXPECT(mpl_block_stack_form_ex(1, 2, 544, 8, 13, 0, 1, 2, 0) < 0);
    (void)mpl_block_stack_form_ex(1, 2, 544, 8, 13, 12, 0, 2, 0);
    EXPECT(mpl_block_stack_last_form() < 0);                                  // nothing was launched by this thread

    // ---- switches and per-device state
    EXPECT(mpl_x3_spin_limit(0) < 0);
    EXPECT(mpl_x3_spin_limit(31) < 0);
    EXPECT(mpl_x3_spin_limit(-1) < 0);
    EXPECT(mpl_x3_spin_limit(23) == MPL_OK);
    for (int m = 0; m < 300; m += 7) (void)mpl_x3_stack_mode(m);
    (void)mpl_x3_stack_mode(0);
    (void)mpl_x3_debug_buffer(nullptr);
    for (int d : {-1, 0, 1, 63, 64, 1000}) {
        (void)mpl_device_error(d);
        (void)mpl_device_error_clear(d);
    }
    {
        float ms[16] = {0};
        int cnt[16] = {0};
        (void)mpl_profile_start();
        (void)mpl_profile_stop(ms, cnt, 16);
        (void)mpl_profile_stop(ms, cnt, 16);      // stop without start
        (void)mpl_profile_stop(nullptr, nullptr, 0);
    }

    // ---- whole-forward entry points: NULL / too small / inconsistent arguments, then well-formed calls without a device.
    // Device pointers are opaque to the host side (never dereferenced there): small host arrays stand in for them.
    std::vector<float> fake(4096, 0.f);
    std::vector<mpl_block_weights> blocks(12);
    std::memset(blocks.data(), 0, blocks.size() * sizeof(mpl_block_weights));
    for (auto& b : blocks) {
        b.ln1_w = b.ln1_b = b.qkv_w = b.qkv_b = b.proj_w = b.proj_b = fake.data();
        b.ln2_w = b.ln2_b = b.fc1_w = b.fc1_b = b.fc2_w = b.fc2_b = fake.data();
    }
    mpl_spt_set set{fake.data(), fake.data(), nullptr, nullptr, fake.data(), blocks.data()};
    mpl_weights w;
    std::memset(&w, 0, sizeof(w));
    w.spt_sets = &set;
    w.spatial_norm_w = w.spatial_norm_b = w.pos_3d_embed = w.pos_3d_view_coding = fake.data();
    w.pos_3d_linear_w = w.pos_3d_linear_b = w.view_norm_w = w.view_norm_b = w.wmean_w = w.wmean_b = fake.data();
    w.head_ln_w = w.head_ln_b = w.head_w = w.head_b = fake.data();
    w.fpt_blocks = blocks.data();
    mpl_inputs in;
    std::memset(&in, 0, sizeof(in));
    in.batch = 8;
    for (int v = 0; v < 4; ++v) in.poses[v] = in.rays[v] = in.centers[v] = fake.data();
    mpl_config cfg{17, 32, 12, 8, 4, 2, MPL_F_POS3D_LEARN, 0};
    const size_t wsb = mpl_forward_workspace_bytes(&cfg, 8);
    std::vector<char> ws(wsb + 256);
    EXPECT(mpl_forward(nullptr, &w, &in, fake.data(), ws.data(), wsb, nullptr) < 0);
    EXPECT(mpl_forward(&cfg, nullptr, &in, fake.data(), ws.data(), wsb, nullptr) < 0);
    EXPECT(mpl_forward(&cfg, &w, nullptr, fake.data(), ws.data(), wsb, nullptr) < 0);
    EXPECT(mpl_forward(&cfg, &w, &in, nullptr, ws.data(), wsb, nullptr) < 0);
    EXPECT(mpl_forward(&cfg, &w, &in, fake.data(), nullptr, wsb, nullptr) < 0);
    EXPECT(mpl_forward(&cfg, &w, &in, fake.data(), ws.data(), wsb / 2, nullptr) < 0);
    {
        mpl_config c2 = cfg;
        c2.num_views = MPL_MAX_VIEWS + 1;
        EXPECT(mpl_forward(&c2, &w, &in, fake.data(), ws.data(), wsb, nullptr) < 0);
        c2 = cfg;
        c2.depth = MPL_MAX_APPS + 3;
        EXPECT(mpl_forward(&c2, &w, &in, fake.data(), ws.data(), wsb, nullptr) < 0);
        c2 = cfg;
        c2.num_joints = 18;
        EXPECT(mpl_forward(&c2, &w, &in, fake.data(), ws.data(), wsb, nullptr) < 0);
        c2 = cfg;
        c2.flags |= MPL_F_KPTOK | MPL_F_RAYS_TOKEN;
        EXPECT(mpl_forward(&c2, &w, &in, fake.data(), ws.data(), wsb, nullptr) < 0);
        mpl_inputs i2 = in;
        i2.batch = -4;
        EXPECT(mpl_forward(&cfg, &w, &i2, fake.data(), ws.data(), wsb, nullptr) < 0);
        i2 = in;
        i2.poses[2] = nullptr;
        EXPECT(mpl_forward(&cfg, &w, &i2, fake.data(), ws.data(), wsb, nullptr) < 0);
    }
    EXPECT(mpl_forward(&cfg, &w, &in, fake.data(), ws.data(), wsb, nullptr) < 0);      // well formed: no device here
    EXPECT(mpl_spt_tokens(&cfg, &w, &in, fake.data(), nullptr) < 0);
    EXPECT(mpl_fuse_head(&cfg, &w, fake.data(), 8, fake.data(), nullptr) < 0);
    EXPECT(mpl_view_fuse(&cfg, &w, fake.data(), 8, fake.data(), nullptr) < 0);
    EXPECT(mpl_view_norm(&cfg, &w, fake.data(), 8, fake.data(), nullptr) < 0);

    // ---- the block stack: schedules of every length, indices beyond the blocks, workspaces of every size
    {
        std::vector<uint8_t> sched(MPL_MAX_APPS + 8);
        for (size_t i = 0; i < sched.size(); ++i) sched[i] = (uint8_t)(i % 12);
        const size_t sb = mpl_block_stack_workspace_bytes(8, 4, 544);
        std::vector<char> sw(sb + 256);
        for (int n_apps : {-1, 0, 1, 13, MPL_MAX_APPS, MPL_MAX_APPS + 1, MPL_MAX_APPS + 8})
            for (size_t bytes : {(size_t)0, sb / 3, sb, sb + 100}) {
                EXPECT(mpl_block_stack(fake.data(), 8, 4, 544, 8, blocks.data(), sched.data(), n_apps, sw.data(), bytes, nullptr) < 0 || n_apps == 0);
                (void)mpl_block_stack_ex(fake.data(), 8, 4, 544, 8, blocks.data(), sched.data(), n_apps, sw.data(), bytes,
                                         MPL_F_NO_SMALL_STACK, nullptr);
            }
        EXPECT(mpl_block_stack(nullptr, 8, 4, 544, 8, blocks.data(), sched.data(), 13, sw.data(), sb, nullptr) < 0);
        EXPECT(mpl_block_stack(fake.data(), 8, 4, 544, 8, nullptr, sched.data(), 13, sw.data(), sb, nullptr) < 0);
        EXPECT(mpl_block_stack(fake.data(), 8, 4, 544, 8, blocks.data(), nullptr, 13, sw.data(), sb, nullptr) < 0);
        EXPECT(mpl_block_stack(fake.data(), 8, 4, 544, 7, blocks.data(), sched.data(), 13, sw.data(), sb, nullptr) < 0);
        EXPECT(mpl_block_stack(fake.data(), 8, 4, 545, 8, blocks.data(), sched.data(), 13, sw.data(), sb, nullptr) < 0);
        EXPECT(mpl_block_stack(fake.data(), 1 << 30, 4, 544, 8, blocks.data(), sched.data(), 13, sw.data(), sb, nullptr) < 0);
        // packed-operand blocks (fp16x2 / bf16 / d32 fields set): the engine selection paths of block_stack_impl
        std::vector<mpl_block_weights> pk = blocks;
        std::vector<uint16_t> op(1 << 16, 0);
        for (auto& b : pk) b.qkv_h2 = b.proj_h2 = b.fc1_h2 = b.fc2_h2 = op.data();
        (void)mpl_block_stack(fake.data(), 64, 4, 544, 8, pk.data(), sched.data(), 13, sw.data(), sb, nullptr);
        for (auto& b : pk) {
            b.qkv_h2 = b.proj_h2 = b.fc1_h2 = b.fc2_h2 = nullptr;
            b.qkv_w16 = b.proj_w16 = b.fc1_w16 = b.fc2_w16 = op.data();
        }
        (void)mpl_block_stack(fake.data(), 64, 4, 544, 8, pk.data(), sched.data(), 13, sw.data(), sb, nullptr);
        pk[3].fc1_w16 = nullptr;          // one block without its operand: the stack must not mix engines
        (void)mpl_block_stack(fake.data(), 64, 4, 544, 8, pk.data(), sched.data(), 13, sw.data(), sb, nullptr);
        std::vector<mpl_block_weights> d32 = blocks;
        for (auto& b : d32) b.qkv_w3 = op.data();
        (void)mpl_block_stack(fake.data(), 4, 68, 32, 8, d32.data(), sched.data(), 3, sw.data(), sb, nullptr);
    }

    // ---- the unit entry points
    EXPECT(mpl_ln_linear(fake.data(), 64, 544, fake.data(), fake.data(), 1e-6f, fake.data(), fake.data(), 544, MPL_EPI_BIAS, nullptr,
                         fake.data(), fake.data(), nullptr) < 0);
    EXPECT(mpl_ln_linear(nullptr, 64, 544, nullptr, nullptr, 1e-6f, fake.data(), fake.data(), 544, 9, nullptr, fake.data(), nullptr, nullptr) < 0);
    EXPECT(mpl_token_attention(fake.data(), 8, 4, 544, 8, fake.data(), nullptr) < 0);
    EXPECT(mpl_token_attention(fake.data(), 8, 4, 544, 7, fake.data(), nullptr) < 0);
    EXPECT(mpl_layernorm(fake.data(), 8, 544, fake.data(), fake.data(), 1e-5f, fake.data(), nullptr) < 0);
    EXPECT(mpl_linear(fake.data(), 51, fake.data(), 544, 8, fake.data(), fake.data(), 1024, nullptr, nullptr, nullptr, nullptr, 1e-5f, 1,
                      fake.data(), nullptr) < 0);
    EXPECT(mpl_linear(nullptr, 0, nullptr, 0, 8, fake.data(), fake.data(), 1024, nullptr, nullptr, nullptr, nullptr, 1e-5f, 0, fake.data(),
                      nullptr) < 0);
    {
        std::vector<uint16_t> dst(64 * 1024, 0);
        EXPECT(mpl_spt_pack(&blocks[0], dst.data(), nullptr) < 0);
        EXPECT(mpl_spt_pack(nullptr, dst.data(), nullptr) < 0);
        EXPECT(mpl_d32_pack(&blocks[0], nullptr, nullptr) < 0);
        EXPECT(mpl_pack_bf16(fake.data(), fake.data(), nullptr, nullptr, 544, 100, dst.data(), nullptr) < 0);
        EXPECT(mpl_pack_bf16(fake.data(), fake.data(), nullptr, nullptr, 544, 544, dst.data(), nullptr) < 0);
        EXPECT(mpl_pack_h2(fake.data(), fake.data(), fake.data(), fake.data(), 544, 2176, dst.data(), nullptr) < 0);
        EXPECT(mpl_pack_h2(fake.data(), fake.data(), nullptr, nullptr, 544, 544, dst.data(), nullptr) < 0);
        EXPECT(mpl_pack_h2_scaled(fake.data(), fake.data(), nullptr, 544, 544, dst.data(), nullptr) < 0);
        EXPECT(mpl_ln_linear_h2(fake.data(), 64, 544, 1, 1e-6f, dst.data(), 544, MPL_EPI_BIAS, nullptr, fake.data(), fake.data(), ws.data(),
                                16, nullptr) < 0);
    }
    {
        float* views[4] = {fake.data(), fake.data(), fake.data(), fake.data()};
        std::vector<double> cams(4 * 16, 0.0);
        EXPECT(mpl_prepare_inputs(fake.data(), nullptr, cams.data(), 8, 4, 17, 1000.f, 1000.f, 1, 0, views, views, views, nullptr) < 0);
        EXPECT(mpl_prepare_inputs(fake.data(), nullptr, cams.data(), 8, MPL_MAX_VIEWS + 1, 17, 1000.f, 1000.f, 1, 0, views, views, views, nullptr) < 0);
        EXPECT(mpl_prepare_inputs(nullptr, nullptr, cams.data(), 8, 4, 17, 1000.f, 1000.f, 1, 0, views, views, views, nullptr) < 0);
        const float sc[3] = {1.f, 2.f, 3.f};
        EXPECT(mpl_pose_metrics(fake.data(), fake.data(), nullptr, 8, 17, sc, sc, fake.data(), nullptr) < 0);
        EXPECT(mpl_pose_metrics_ex(fake.data(), fake.data(), fake.data(), 8, 17, nullptr, nullptr, 0x1ffffu, fake.data(), nullptr) < 0);
        EXPECT(mpl_pose_metrics(nullptr, fake.data(), nullptr, 8, 17, sc, sc, fake.data(), nullptr) < 0);
        EXPECT(mpl_pose_metrics(fake.data(), fake.data(), nullptr, 8, 40, sc, sc, fake.data(), nullptr) < 0);
    }
    std::printf("host_driver: %d expectation(s) failed\n", failures);
    return failures ? 1 : 0;
}
