// v0_core_ext.cpp -- compiled binding of the `v0_core` operator surface (PyBind11 + torch, built with g++).
//
// The reference's `v0_core` is a PyBind11 extension (v0/src/bindings/module.cpp:874-1482) whose operators take and
// return torch tensors.  The product boundary of this repository is the C ABI of include/liuzhou_hip.h; this file is the
// thin compiled layer a maintainer would put where the reference has module.cpp: per operator it checks the arguments,
// allocates the outputs with torch, and calls the C ABI with `tensor.data_ptr()` and the current HIP stream -- nothing
// else happens here (no kernels, no rules).  `liuzhou_amd/v0_core.py` is the same layer over ctypes (12 - 22 us of host
// time per call); this one costs a few microseconds.  Two builds of the ABI are bound, chosen by the tensors' device
// exactly as the reference extension dispatches (v0/src/game/fast_legal_mask.cpp:453): libliuzhou_hip.so for HIP tensors,
// libliuzhou_host.so for CPU tensors.  There is no fallback between them.
#include <dlfcn.h>
#include <torch/extension.h>

#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>

#include <array>
#include <stdexcept>
#include <string>
#include <tuple>

#include "../../include/liuzhou_hip.h"

namespace {

using at::Tensor;

struct Abi {
    void* handle = nullptr;
    std::string path;
    template <class Fn>
    Fn get(const char* name) const {
        if (!handle) throw std::runtime_error("liuzhou_amd: library " + path + " is not loaded (no fallback)");
        void* p = dlsym(handle, name);
        if (!p) throw std::runtime_error(std::string("liuzhou_amd: ") + path + " does not export " + name);
        return reinterpret_cast<Fn>(p);
    }
};
Abi g_hip, g_host;

void bind_libraries(const std::string& hip_path, const std::string& host_path) {
    auto open = [](Abi& a, const std::string& p) {
        if (p.empty()) return;
        void* h = dlopen(p.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) throw std::runtime_error("liuzhou_amd: cannot load " + p + ": " + dlerror());
        a.handle = h;
        a.path = p;
    };
    open(g_hip, hip_path);
    open(g_host, host_path);
}

const Abi& abi_for(const Tensor& t) { return t.is_cuda() ? g_hip : g_host; }
#define LZ_FN(t, name) abi_for(t).get<decltype(&name)>(#name)

void* stream_of(const Tensor& t) {
    if (!t.is_cuda()) return nullptr;                        // host build: no stream
    return reinterpret_cast<void*>(c10::hip::getCurrentHIPStream(t.device().index()).stream());
}
// current device = the tensors' device for the launch (torch.cuda.device(dev) of the ctypes layer)
struct DeviceScope {
    c10::hip::OptionalHIPGuard guard;
    explicit DeviceScope(const Tensor& t) { if (t.is_cuda()) guard.set_index(t.device().index()); }
};

const char* status_text(int s) {
    switch (s) {
        case 0: return "ok";
        case -1: return "invalid argument";
        case -2: return "unsupported dimensions";
        case -3: return "kernel launch failed";
        case -4: return "misaligned pointer";
        case -5: return "illegal action for the state";
        default: return "unknown status";
    }
}
void check(int status, const char* op) {
    if (status != 0)
        throw std::runtime_error(std::string("liuzhou_amd.") + op + " failed: " + status_text(status) + " (" +
                                 std::to_string(status) + ")");
}

Tensor as(const Tensor& t, at::ScalarType dt) { return (t.scalar_type() == dt ? t : t.to(dt)).contiguous(); }
template <class T> T* ptr(const Tensor& t) { return t.defined() ? reinterpret_cast<T*>(t.data_ptr()) : nullptr; }

struct States {
    std::array<Tensor, 12> t;
    LzStateSoA soa() const {
        LzStateSoA s;
        s.board = ptr<int8_t>(t[0]); s.marks_black = ptr<uint8_t>(t[1]); s.marks_white = ptr<uint8_t>(t[2]);
        s.phase = ptr<int64_t>(t[3]); s.current_player = ptr<int64_t>(t[4]);
        s.pending_marks_required = ptr<int64_t>(t[5]); s.pending_marks_remaining = ptr<int64_t>(t[6]);
        s.pending_captures_required = ptr<int64_t>(t[7]); s.pending_captures_remaining = ptr<int64_t>(t[8]);
        s.forced_removals_done = ptr<int64_t>(t[9]); s.move_count = ptr<int64_t>(t[10]);
        s.moves_since_capture = ptr<int64_t>(t[11]);
        return s;
    }
};
// the ten / twelve state tensors in the dtypes of the ABI (int8 board, bool marks, int64 scalars); the two counters
// default to `phase` where an operator does not read them
States pack_states(const Tensor& board, const Tensor& mb, const Tensor& mw, const Tensor& phase, const Tensor& player,
                   const Tensor& pmr, const Tensor& pmm, const Tensor& pcr, const Tensor& pcm, const Tensor& forced,
                   const Tensor* move_count = nullptr, const Tensor* msc = nullptr) {
    States s;
    s.t[0] = as(board, at::kChar); s.t[1] = as(mb, at::kBool); s.t[2] = as(mw, at::kBool);
    const Tensor* rest[7] = {&phase, &player, &pmr, &pmm, &pcr, &pcm, &forced};
    for (int i = 0; i < 7; ++i) s.t[3 + i] = as(*rest[i], at::kLong);
    s.t[10] = move_count ? as(*move_count, at::kLong) : s.t[3];
    s.t[11] = msc ? as(*msc, at::kLong) : s.t[3];
    return s;
}
at::TensorOptions opts(const Tensor& like, at::ScalarType dt) { return at::TensorOptions().dtype(dt).device(like.device()); }

// ---- module.cpp:1294-1310 -------------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor> encode_actions_fast(const Tensor& board, const Tensor& mb, const Tensor& mw, const Tensor& phase,
                                               const Tensor& player, const Tensor& pmr, const Tensor& pmm, const Tensor& pcr,
                                               const Tensor& pcm, const Tensor& forced, int64_t placement_dim,
                                               int64_t movement_dim, int64_t selection_dim, int64_t auxiliary_dim) {
    const States st = pack_states(board, mb, mw, phase, player, pmr, pmm, pcr, pcm, forced);
    const int64_t B = st.t[0].size(0), T = placement_dim + movement_dim + selection_dim + auxiliary_dim;
    Tensor mask = at::empty({B, T}, opts(board, at::kBool)), meta = at::empty({B, T, 4}, opts(board, at::kInt));
    const LzStateSoA s = st.soa();
    DeviceScope scope(board);
    check(LZ_FN(board, lz_encode_actions_fast)(&s, B, placement_dim, movement_dim, selection_dim, auxiliary_dim,
                                               ptr<uint8_t>(mask), ptr<int32_t>(meta), stream_of(board)),
          "encode_actions_fast");
    return {mask, meta};
}

std::array<Tensor, 12> alloc_states(int64_t n, const Tensor& like) {
    std::array<Tensor, 12> o;
    o[0] = at::empty({n, 6, 6}, opts(like, at::kChar));
    o[1] = at::empty({n, 6, 6}, opts(like, at::kBool));
    o[2] = at::empty({n, 6, 6}, opts(like, at::kBool));
    for (int i = 3; i < 12; ++i) o[i] = at::empty({n}, opts(like, at::kLong));
    return o;
}

// ---- module.cpp:1311-1327 -------------------------------------------------------------------------------------------
std::vector<Tensor> batch_apply_moves(const Tensor& board, const Tensor& mb, const Tensor& mw, const Tensor& phase,
                                      const Tensor& player, const Tensor& pmr, const Tensor& pmm, const Tensor& pcr,
                                      const Tensor& pcm, const Tensor& forced, const Tensor& move_count, const Tensor& msc,
                                      const Tensor& action_codes, const Tensor& parent_indices) {
    const States st = pack_states(board, mb, mw, phase, player, pmr, pmm, pcr, pcm, forced, &move_count, &msc);
    const Tensor codes = as(action_codes.to(board.device()), at::kInt);
    const Tensor parents = as(parent_indices.to(board.device()), at::kLong).view({-1});
    if (codes.dim() != 2 || codes.size(1) != 4) throw std::runtime_error("action_codes must be (N, 4).");
    const int64_t N = codes.size(0);
    if (parents.numel() != N) throw std::runtime_error("parent_indices must align with action_codes.");
    States out;
    out.t = alloc_states(N, board);
    const LzStateSoA si = st.soa(), so = out.soa();
    DeviceScope scope(board);
    check(LZ_FN(board, lz_batch_apply_moves)(&si, st.t[0].size(0), ptr<int32_t>(codes), ptr<int64_t>(parents), N, &so,
                                             stream_of(board)),
          "batch_apply_moves");
    return std::vector<Tensor>(out.t.begin(), out.t.end());
}

// The in-place operators hand the caller's storages straight to the kernels (nothing is converted: a copy would not be
// mutated), so what the non-in-place operators get from as() has to be CHECKED here: dtype (int8 board, bool marks,
// int64 scalars), one device, contiguity, one batch size -- as module.cpp's TORCH_CHECKs do; an int32 `phase` or a CPU
// tensor among HIP ones would otherwise be an out-of-bounds or garbage device access.
void check_inplace_states(const Tensor* const (&all)[12], const char* op) {
    static const char* names[12] = {"board", "marks_black", "marks_white", "phase", "current_player",
                                    "pending_marks_required", "pending_marks_remaining", "pending_captures_required",
                                    "pending_captures_remaining", "forced_removals_done", "move_count",
                                    "moves_since_capture"};
    const Tensor& board = *all[0];
    if (board.dim() < 1) throw std::runtime_error(std::string(op) + ": board must be [B, 6, 6]");
    const int64_t B = board.size(0);
    for (int i = 0; i < 12; ++i) {
        const Tensor& t = *all[i];
        const at::ScalarType want = i == 0 ? at::kChar : i < 3 ? at::kBool : at::kLong;
        if (!t.defined() || t.scalar_type() != want)
            throw std::runtime_error(std::string(op) + ": " + names[i] + " must be " +
                                     (i == 0 ? "int8" : i < 3 ? "bool" : "int64") + " (it is mutated in place)");
        if (t.device() != board.device())
            throw std::runtime_error(std::string(op) + ": " + names[i] + " is not on the board's device");
        if (!t.is_contiguous())
            throw std::runtime_error(std::string(op) + ": state tensors must be contiguous (they are mutated)");
        if (t.numel() != (i < 3 ? B * 36 : B))
            throw std::runtime_error(std::string(op) + ": " + names[i] + " does not hold " + std::to_string(B) + " states");
    }
}

// ---- fast_apply_moves_cuda.cu:746-917 -------------------------------------------------------------------------------
void batch_apply_moves_inplace(const Tensor& board, const Tensor& mb, const Tensor& mw, const Tensor& phase,
                               const Tensor& player, const Tensor& pmr, const Tensor& pmm, const Tensor& pcr,
                               const Tensor& pcm, const Tensor& forced, const Tensor& move_count, const Tensor& msc,
                               const Tensor& action_codes, const Tensor& slot_indices) {
    States st;
    const Tensor* all[12] = {&board, &mb, &mw, &phase, &player, &pmr, &pmm, &pcr, &pcm, &forced, &move_count, &msc};
    check_inplace_states(all, "batch_apply_moves_inplace");
    for (int i = 0; i < 12; ++i) st.t[i] = *all[i];
    const Tensor codes = as(action_codes.to(board.device()), at::kInt);
    const Tensor slots = as(slot_indices.to(board.device()), at::kLong).view({-1});
    if (codes.dim() != 2 || codes.size(1) != 4 || codes.size(0) != slots.numel())
        throw std::runtime_error("batch_apply_moves_inplace: action_codes must be [N, 4] with one row per slot index");
    const LzStateSoA s = st.soa();
    DeviceScope scope(board);
    check(LZ_FN(board, lz_batch_apply_moves_inplace)(&s, board.size(0), ptr<int32_t>(codes), ptr<int64_t>(slots),
                                                     slots.numel(), stream_of(board)),
          "batch_apply_moves_inplace");
}

// ---- module.cpp:1286-1293 -------------------------------------------------------------------------------------------
Tensor states_to_model_input(const Tensor& board, const Tensor& mb, const Tensor& mw, const Tensor& phase,
                             const Tensor& player) {
    const Tensor b = as(board, at::kChar), m1 = as(mb, at::kBool), m2 = as(mw, at::kBool), ph = as(phase, at::kLong),
                 cp = as(player, at::kLong);
    const int64_t B = b.size(0);
    Tensor out = at::empty({B, 11, 6, 6}, opts(board, at::kFloat));
    DeviceScope scope(board);
    check(LZ_FN(board, lz_states_to_model_input)(ptr<int8_t>(b), ptr<uint8_t>(m1), ptr<uint8_t>(m2), ptr<int64_t>(ph),
                                                 ptr<int64_t>(cp), B, ptr<float>(out), stream_of(board)),
          "states_to_model_input");
    return out;
}

// ---- module.cpp:1328-1338 -------------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor> project_policy_logits_fast(const Tensor& log_p1, const Tensor& log_p2, const Tensor& log_pmc,
                                                      const Tensor& legal_mask, int64_t placement_dim, int64_t movement_dim,
                                                      int64_t selection_dim, int64_t auxiliary_dim) {
    if (legal_mask.scalar_type() != at::kBool) throw std::runtime_error("legal_mask must be of dtype bool.");
    if (log_p1.scalar_type() != log_p2.scalar_type() || log_p1.scalar_type() != log_pmc.scalar_type())
        throw std::runtime_error("All policy heads must share the same dtype.");
    const auto out_dtype = log_p1.scalar_type();
    const int64_t B = log_p1.size(0), T = placement_dim + movement_dim + selection_dim + auxiliary_dim;
    const Tensor p1 = as(log_p1.reshape({B, -1}), at::kFloat), p2 = as(log_p2.reshape({B, -1}), at::kFloat),
                 pm = as(log_pmc.reshape({B, -1}), at::kFloat);
    if (legal_mask.dim() != 2 || legal_mask.size(0) != B || legal_mask.size(1) != T)
        throw std::runtime_error("legal_mask expected shape (" + std::to_string(B) + ", " + std::to_string(T) + ").");
    const Tensor mk = legal_mask.contiguous();
    Tensor probs = at::empty({B, T}, opts(log_p1, at::kFloat)), ml = at::empty({B, T}, opts(log_p1, at::kFloat));
    DeviceScope scope(log_p1);
    check(LZ_FN(log_p1, lz_project_policy_logits_fast)(ptr<float>(p1), ptr<float>(p2), ptr<float>(pm), ptr<uint8_t>(mk), B,
                                                       placement_dim, movement_dim, selection_dim, auxiliary_dim,
                                                       ptr<float>(probs), ptr<float>(ml), stream_of(log_p1)),
          "project_policy_logits_fast");
    if (out_dtype != at::kFloat) return {probs.to(out_dtype), ml.to(out_dtype)};
    return {probs, ml};
}

constexpr int64_t kPackCap = 80;   // >= max legal actions of any state (placement 36, movement <= 72)

std::tuple<Tensor, Tensor, Tensor, Tensor> root_pack_rows(const Tensor& legal_mask, const Tensor& probs,
                                                          const Tensor& metadata, int64_t cap) {
    const Tensor mk = as(legal_mask, at::kBool), pr = as(probs, at::kFloat), md = as(metadata, at::kInt);
    const int64_t B = mk.size(0), T = mk.size(1);
    Tensor counts = at::empty({B}, opts(mk, at::kInt)), lidx = at::empty({B, cap}, opts(mk, at::kInt)),
           pri = at::empty({B, cap}, opts(mk, at::kFloat)), codes = at::empty({B, cap, 4}, opts(mk, at::kInt));
    DeviceScope scope(mk);
    check(LZ_FN(mk, lz_root_pack_rows)(ptr<uint8_t>(mk), ptr<float>(pr), ptr<int32_t>(md), B, T, cap, ptr<int32_t>(counts),
                                       ptr<int32_t>(lidx), ptr<float>(pri), ptr<int32_t>(codes), stream_of(mk)),
          "root_pack_sparse_actions");
    return {counts, lidx, pri, codes};
}

// ---- module.cpp:1357-1362: the reference's 10-tuple; one host read of {R, Amax, N} ------------------------------------
std::vector<Tensor> root_pack_sparse_actions(const Tensor& legal_mask, const Tensor& probs, const Tensor& metadata) {
    if (legal_mask.dim() != 2 || probs.dim() != 2 || metadata.dim() != 3 || metadata.size(2) != 4)
        throw std::runtime_error("legal_mask [B,A], probs [B,A], metadata [B,A,4] expected");
    auto [counts, lidx, pri, codes] = root_pack_rows(legal_mask, probs, metadata, kPackCap);
    const int64_t B = counts.size(0);
    Tensor rank = at::empty({B}, opts(counts, at::kInt)), child_off = at::empty({B}, opts(counts, at::kLong)),
           sizes = at::zeros({3}, opts(counts, at::kLong));
    DeviceScope scope(counts);
    check(LZ_FN(counts, lz_root_pack_plan)(ptr<int32_t>(counts), B, ptr<int32_t>(rank), ptr<int64_t>(child_off),
                                           ptr<int64_t>(sizes), stream_of(counts)),
          "root_pack_sparse_actions");
    const Tensor host = sizes.cpu();                              // the one host synchronisation
    const int64_t R = host[0].item<int64_t>(), N = host[2].item<int64_t>();
    const int64_t M = R == 0 ? 0 : host[1].item<int64_t>();
    auto e = [&](at::IntArrayRef shape, at::ScalarType dt) { return at::empty(shape, opts(counts, dt)); };
    Tensor terminal_mask = e({B}, at::kBool), valid_root_indices = e({R}, at::kLong), counts_v = e({R}, at::kLong),
           valid_mask = e({R, M}, at::kBool), legal_index_mat = e({R, M}, at::kLong), priors_mat = e({R, M}, at::kFloat),
           action_code_mat = e({R, M, 4}, at::kInt), pack_flat_idx = e({N}, at::kLong), action_codes_all = e({N, 4}, at::kInt),
           parent_indices_all = e({N}, at::kLong);
    check(LZ_FN(counts, lz_root_pack_fill)(
              ptr<int32_t>(counts), ptr<int32_t>(lidx), ptr<float>(pri), ptr<int32_t>(codes), ptr<int32_t>(rank),
              ptr<int64_t>(child_off), B, lidx.size(1), R, M, N, ptr<uint8_t>(terminal_mask), ptr<int64_t>(valid_root_indices),
              ptr<int64_t>(counts_v), ptr<uint8_t>(valid_mask), ptr<int64_t>(legal_index_mat), ptr<float>(priors_mat),
              ptr<int32_t>(action_code_mat), ptr<int64_t>(pack_flat_idx), ptr<int32_t>(action_codes_all),
              ptr<int64_t>(parent_indices_all), stream_of(counts)),
          "root_pack_sparse_actions");
    return {terminal_mask, valid_root_indices, counts_v, valid_mask, legal_index_mat, priors_mat, action_code_mat,
            pack_flat_idx, action_codes_all, parent_indices_all};
}

// ---- module.cpp:1349-1356 -------------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor, Tensor> root_puct_allocate_visits(const Tensor& priors, const Tensor& leaf_values,
                                                             const Tensor& valid_mask, int64_t num_simulations,
                                                             double exploration_weight) {
    if (priors.dim() != 2 || leaf_values.dim() != 2 || valid_mask.dim() != 2)
        throw std::runtime_error("priors / leaf_values / valid_mask must be 2D [R, A]");
    if (priors.sizes() != leaf_values.sizes() || priors.sizes() != valid_mask.sizes())
        throw std::runtime_error("priors, leaf_values and valid_mask shape mismatch");
    if (num_simulations <= 0) throw std::runtime_error("num_simulations must be positive");
    const Tensor p = as(priors, at::kFloat), lv = as(leaf_values, at::kFloat), vm = as(valid_mask, at::kBool);
    const int64_t R = p.size(0), A = p.size(1);
    Tensor visits = at::zeros({R, A}, opts(p, at::kFloat)), vs = at::zeros({R, A}, opts(p, at::kFloat)),
           rv = at::zeros({R}, opts(p, at::kFloat));
    if (R == 0 || A == 0) return {visits, vs, rv};
    DeviceScope scope(p);
    check(LZ_FN(p, lz_root_puct_allocate_visits)(ptr<float>(p), ptr<float>(lv), ptr<uint8_t>(vm), R, A, num_simulations,
                                                 (float)exploration_weight, ptr<float>(visits), ptr<float>(vs),
                                                 ptr<float>(rv), stream_of(p)),
          "root_puct_allocate_visits");
    return {visits, vs, rv};
}

// ---- module.cpp:1374-1386 -------------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> root_finalize_from_visits(
    const Tensor& legal_index_mat, const Tensor& action_code_mat, const Tensor& valid_mask, const Tensor& visits,
    const Tensor& value_sum, const Tensor& valid_root_indices, int64_t batch_size, int64_t total_action_dim,
    const Tensor& root_temperatures, bool sample_moves, const c10::optional<Tensor>& uniforms) {
    const Tensor li = as(legal_index_mat, at::kLong), ac = as(action_code_mat, at::kInt), vm = as(valid_mask, at::kBool),
                 vi = as(visits, at::kFloat), vs = as(value_sum, at::kFloat), roots = as(valid_root_indices, at::kLong),
                 temps = as(root_temperatures, at::kFloat);
    const int64_t R = li.size(0), M = li.dim() == 2 ? li.size(1) : 0, B = batch_size, T = total_action_dim;
    Tensor policy = at::empty({B, T}, opts(vi, at::kFloat)), cidx = at::empty({B}, opts(vi, at::kLong)),
           ccodes = at::empty({B, 4}, opts(vi, at::kInt)), cvalid = at::empty({B}, opts(vi, at::kBool)),
           rv = at::empty({R}, opts(vi, at::kFloat));
    Tensor u;
    if (sample_moves && M > 1)
        u = uniforms.has_value() && uniforms->defined() ? as(*uniforms, at::kFloat) : at::rand({R}, opts(vi, at::kFloat));
    DeviceScope scope(vi);
    check(LZ_FN(vi, lz_root_finalize_from_visits)(
              ptr<int64_t>(li), ptr<int32_t>(ac), ptr<uint8_t>(vm), ptr<float>(vi), ptr<float>(vs), ptr<int64_t>(roots), R, M,
              B, T, ptr<float>(temps), u.defined() ? ptr<float>(u) : nullptr, ptr<float>(policy), ptr<int64_t>(cidx),
              ptr<int32_t>(ccodes), ptr<uint8_t>(cvalid), ptr<float>(rv), stream_of(vi)),
          "root_finalize_from_visits");
    return {policy, cidx, ccodes, cvalid, rv};
}

// ---- module.cpp:1387-1409 -------------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor, Tensor> self_play_step_inplace(
    const Tensor& board, const Tensor& mb, const Tensor& mw, const Tensor& phase, const Tensor& player, const Tensor& pmr,
    const Tensor& pmm, const Tensor& pcr, const Tensor& pcm, const Tensor& forced, const Tensor& move_count, const Tensor& msc,
    const Tensor& plies, const Tensor& done, const Tensor& active_idx, const Tensor& chosen_action_codes,
    const Tensor& terminal_mask, const Tensor& chosen_valid_mask, int64_t max_game_plies, double soft_value_k) {
    if (max_game_plies <= 0) throw std::runtime_error("max_game_plies must be positive");
    if (done.scalar_type() != at::kBool || plies.scalar_type() != at::kLong)
        throw std::runtime_error("plies must be int64 and done must be bool");
    States st;
    const Tensor* all[12] = {&board, &mb, &mw, &phase, &player, &pmr, &pmm, &pcr, &pcm, &forced, &move_count, &msc};
    check_inplace_states(all, "self_play_step_inplace");
    for (int i = 0; i < 12; ++i) st.t[i] = *all[i];
    if (!plies.is_contiguous() || !done.is_contiguous())
        throw std::runtime_error("self_play_step_inplace: state tensors must be contiguous (they are mutated)");
    if (plies.device() != board.device() || done.device() != board.device() || plies.numel() != board.size(0) ||
        done.numel() != board.size(0))
        throw std::runtime_error("self_play_step_inplace: plies / done must hold one entry per state on the board's device");
    const Tensor act = as(active_idx.to(board.device()), at::kLong).view({-1});
    const Tensor codes = as(chosen_action_codes.to(board.device()), at::kInt);
    const Tensor term = as(terminal_mask.to(board.device()), at::kBool).view({-1});
    const Tensor cval = as(chosen_valid_mask.to(board.device()), at::kBool).view({-1});
    const int64_t A = act.numel();
    if (codes.dim() != 2 || codes.size(0) != A || codes.size(1) != 4) throw std::runtime_error("chosen_action_codes must be [A, 4]");
    if (term.numel() != A || cval.numel() != A) throw std::runtime_error("terminal_mask / chosen_valid_mask batch mismatch");
    if (A == 0) {
        Tensor e = at::empty({0}, opts(board, at::kFloat));
        return {at::empty({0}, opts(board, at::kLong)), e, e.clone()};
    }
    Tensor kind = at::empty({A}, opts(board, at::kInt)), res = at::empty({A}, opts(board, at::kFloat)),
           soft = at::empty({A}, opts(board, at::kFloat));
    const LzStateSoA s = st.soa();
    {
        DeviceScope scope(board);
        check(LZ_FN(board, lz_self_play_step_inplace)(&s, board.size(0), ptr<int64_t>(plies), ptr<uint8_t>(done),
                                                      ptr<int64_t>(act), A, ptr<int32_t>(codes), ptr<uint8_t>(term),
                                                      ptr<uint8_t>(cval), max_game_plies, (float)soft_value_k,
                                                      ptr<int32_t>(kind), ptr<float>(res), ptr<float>(soft), stream_of(board)),
              "self_play_step_inplace");
    }
    // games ended before the move, then games ended by it, each in active order (module.cpp:724-741, :838-856)
    const Tensor order = at::cat({at::nonzero(kind.eq(1)).view({-1}), at::nonzero(kind.eq(2)).view({-1})});
    return {act.index_select(0, order), res.index_select(0, order), soft.index_select(0, order)};
}

// ---- module.cpp:1410-1420 -------------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor, Tensor> finalize_trajectory_inplace(const Tensor& value_targets, const Tensor& soft_value_targets,
                                                               const Tensor& player_signs, const Tensor& step_index_matrix,
                                                               const Tensor& step_counts, const Tensor& slots,
                                                               const Tensor& result_from_black,
                                                               const Tensor& soft_value_from_black) {
    if (!value_targets.is_contiguous() || !soft_value_targets.is_contiguous())
        throw std::runtime_error("target buffers must be contiguous");
    if (value_targets.scalar_type() != at::kFloat || soft_value_targets.scalar_type() != at::kFloat)
        throw std::runtime_error("target buffers must be float32");
    Tensor counts_out = at::zeros({3}, opts(value_targets, at::kLong));
    const Tensor sl = as(slots.to(value_targets.device()), at::kLong).view({-1});
    const int64_t F = sl.numel();
    if (F == 0) {
        Tensor e = at::empty({0}, opts(value_targets, at::kLong));
        return {e, e.clone(), counts_out};
    }
    const Tensor res = as(result_from_black.to(value_targets.device()), at::kFloat).view({-1});
    const Tensor sft = as(soft_value_from_black.to(value_targets.device()), at::kFloat).view({-1});
    if (res.numel() != F || sft.numel() != F)
        throw std::runtime_error("result_from_black / soft_value_from_black must align with slots");
    const Tensor signs = as(player_signs, at::kChar), sim = as(step_index_matrix, at::kLong), sc = as(step_counts, at::kLong);
    Tensor keep = at::empty({F}, opts(value_targets, at::kBool)), fcounts = at::empty({F}, opts(value_targets, at::kLong));
    {
        DeviceScope scope(value_targets);
        check(LZ_FN(value_targets, lz_finalize_trajectory_inplace)(
                  ptr<float>(value_targets), ptr<float>(soft_value_targets), ptr<int8_t>(signs), ptr<int64_t>(sim),
                  ptr<int64_t>(sc), sim.size(0), sim.size(1), ptr<int64_t>(sl), ptr<float>(res), ptr<float>(sft), F,
                  ptr<uint8_t>(keep), ptr<int64_t>(fcounts), ptr<int64_t>(counts_out), stream_of(value_targets)),
              "finalize_trajectory_inplace");
    }
    const Tensor kidx = at::nonzero(keep).view({-1});
    return {sl.index_select(0, kidx), fcounts.index_select(0, kidx), counts_out};
}

// ---- the operators the reference defines as tensor-library compositions (no kernel of their own) --------------------
// root_sparse_writeback (module.cpp:365-439): scatter the packed per-root policy back to dense rows and pick the codes
std::tuple<Tensor, Tensor, Tensor, Tensor> root_sparse_writeback(const Tensor& legal_index_mat, const Tensor& action_code_mat,
                                                                 const Tensor& valid_mask, const Tensor& legal_policy,
                                                                 const Tensor& local_picks, const Tensor& valid_root_indices,
                                                                 int64_t batch_size, int64_t total_action_dim) {
    if (batch_size < 0) throw std::runtime_error("batch_size must be non-negative");
    if (total_action_dim <= 0) throw std::runtime_error("total_action_dim must be positive");
    if (legal_index_mat.dim() != 2 || valid_mask.dim() != 2 || legal_policy.dim() != 2)
        throw std::runtime_error("legal_index_mat / valid_mask / legal_policy must be [R, M]");
    if (action_code_mat.dim() != 3 || action_code_mat.size(2) != 4) throw std::runtime_error("action_code_mat must be [R, M, 4]");
    const int64_t R = legal_index_mat.size(0), M = legal_index_mat.size(1);
    if (valid_mask.size(0) != R || valid_mask.size(1) != M || legal_policy.size(0) != R || legal_policy.size(1) != M ||
        action_code_mat.size(0) != R || action_code_mat.size(1) != M)
        throw std::runtime_error("legal_index_mat / valid_mask / legal_policy / action_code_mat shape mismatch");
    if (local_picks.dim() != 1 || local_picks.size(0) != R || valid_root_indices.dim() != 1 || valid_root_indices.size(0) != R)
        throw std::runtime_error("local_picks / valid_root_indices must be [R]");
    const auto dev = legal_index_mat.device();
    for (const Tensor* t : {&action_code_mat, &valid_mask, &legal_policy, &local_picks, &valid_root_indices})
        if (t->device() != dev) throw std::runtime_error("all tensors must be on the same device");
    const Tensor idx = as(legal_index_mat, at::kLong), codes = as(action_code_mat, at::kInt);
    const Tensor weights = as(legal_policy, at::kFloat) * as(valid_mask, at::kBool).to(at::kFloat);
    const Tensor picks = as(local_picks, at::kLong), roots = as(valid_root_indices, at::kLong);
    Tensor rows = at::zeros({R, total_action_dim}, opts(idx, at::kFloat));
    rows.scatter_add_(1, idx, weights);
    Tensor policy_dense = at::zeros({batch_size, total_action_dim}, opts(idx, at::kFloat));
    Tensor chosen_idx = at::full({batch_size}, -1, opts(idx, at::kLong));
    Tensor chosen_codes = at::full({batch_size, 4}, -1, opts(idx, at::kInt));
    Tensor chosen_valid = at::zeros({batch_size}, opts(idx, at::kBool));
    policy_dense.index_copy_(0, roots, rows);
    chosen_idx.index_copy_(0, roots, idx.gather(1, picks.view({-1, 1})).view({-1}));
    chosen_codes.index_copy_(0, roots, codes.gather(1, picks.view({-1, 1, 1}).expand({-1, 1, 4})).view({-1, 4}));
    chosen_valid.index_fill_(0, roots, true);
    return {policy_dense, chosen_idx, chosen_codes, chosen_valid};
}

// postprocess_value_head (v0/src/net/encoding.cpp:81-89): WDL logits -> P(win) - P(loss), scalar head -> tanh
Tensor postprocess_value_head(const Tensor& raw_values) {
    if (raw_values.dim() >= 2 && raw_values.size(-1) == 3) {
        const Tensor p = at::softmax(raw_values, -1);
        return p.select(-1, 0) - p.select(-1, 2);
    }
    return at::tanh(raw_values);
}

// apply_temperature_scaling (v0/src/net/encoding.cpp:91-113): p^(1/T) over the positive entries, renormalised along dim
Tensor apply_temperature_scaling(const Tensor& probs_in, double temperature, int64_t dim) {
    const Tensor probs = probs_in.contiguous();
    if (temperature <= 1e-6) return probs.clone();
    const int64_t d = dim < 0 ? dim + probs.dim() : dim;
    if (d < 0 || d >= probs.dim()) throw std::runtime_error("Invalid dimension for temperature scaling");
    const Tensor powered = at::where(probs > 0, probs.pow(1.0 / std::max(temperature, 1e-6)), at::zeros_like(probs));
    const Tensor sums = powered.sum(d, /*keepdim=*/true);
    return at::where(sums > 0, powered / sums, at::zeros_like(powered));
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    namespace py = pybind11;
    m.doc() = "compiled v0_core operator layer over the C ABI of include/liuzhou_hip.h";
    m.def("bind_libraries", &bind_libraries, py::arg("hip_library"), py::arg("host_library"));
    m.def("encode_actions_fast", &encode_actions_fast, py::arg("board"), py::arg("marks_black"), py::arg("marks_white"),
          py::arg("phase"), py::arg("current_player"), py::arg("pending_marks_required"), py::arg("pending_marks_remaining"),
          py::arg("pending_captures_required"), py::arg("pending_captures_remaining"), py::arg("forced_removals_done"),
          py::arg("placement_dim"), py::arg("movement_dim"), py::arg("selection_dim"), py::arg("auxiliary_dim"));
    m.def("batch_apply_moves", [](const Tensor& a, const Tensor& b, const Tensor& c, const Tensor& d, const Tensor& e,
                                  const Tensor& f, const Tensor& g, const Tensor& h, const Tensor& i, const Tensor& j,
                                  const Tensor& k, const Tensor& l, const Tensor& codes, const Tensor& parents) {
              const std::vector<Tensor> v = batch_apply_moves(a, b, c, d, e, f, g, h, i, j, k, l, codes, parents);
              py::tuple t(v.size());
              for (size_t n = 0; n < v.size(); ++n) t[n] = v[n];
              return t;
          },
          py::arg("board"), py::arg("marks_black"), py::arg("marks_white"), py::arg("phase"), py::arg("current_player"),
          py::arg("pending_marks_required"), py::arg("pending_marks_remaining"), py::arg("pending_captures_required"),
          py::arg("pending_captures_remaining"), py::arg("forced_removals_done"), py::arg("move_count"),
          py::arg("moves_since_capture"), py::arg("action_codes"), py::arg("parent_indices"));
    m.def("batch_apply_moves_inplace", &batch_apply_moves_inplace, py::arg("board"), py::arg("marks_black"),
          py::arg("marks_white"), py::arg("phase"), py::arg("current_player"), py::arg("pending_marks_required"),
          py::arg("pending_marks_remaining"), py::arg("pending_captures_required"), py::arg("pending_captures_remaining"),
          py::arg("forced_removals_done"), py::arg("move_count"), py::arg("moves_since_capture"), py::arg("action_codes"),
          py::arg("slot_indices"));
    m.def("states_to_model_input", &states_to_model_input, py::arg("board"), py::arg("marks_black"), py::arg("marks_white"),
          py::arg("phase"), py::arg("current_player"));
    m.def("project_policy_logits_fast", &project_policy_logits_fast, py::arg("log_p1"), py::arg("log_p2"), py::arg("log_pmc"),
          py::arg("legal_mask"), py::arg("placement_dim"), py::arg("movement_dim"), py::arg("selection_dim"),
          py::arg("auxiliary_dim"));
    m.def("root_pack_rows", &root_pack_rows, py::arg("legal_mask"), py::arg("probs"), py::arg("metadata"),
          py::arg("cap") = kPackCap);
    m.def("root_pack_sparse_actions", [](const Tensor& a, const Tensor& b, const Tensor& c) {
              const std::vector<Tensor> v = root_pack_sparse_actions(a, b, c);
              py::tuple t(v.size());
              for (size_t n = 0; n < v.size(); ++n) t[n] = v[n];
              return t;
          },
          py::arg("legal_mask"), py::arg("probs"), py::arg("metadata"));
    m.def("root_puct_allocate_visits", &root_puct_allocate_visits, py::arg("priors"), py::arg("leaf_values"),
          py::arg("valid_mask"), py::arg("num_simulations"), py::arg("exploration_weight"));
    m.def("root_finalize_from_visits", &root_finalize_from_visits, py::arg("legal_index_mat"), py::arg("action_code_mat"),
          py::arg("valid_mask"), py::arg("visits"), py::arg("value_sum"), py::arg("valid_root_indices"), py::arg("batch_size"),
          py::arg("total_action_dim"), py::arg("root_temperatures"), py::arg("sample_moves"),
          py::arg("uniforms") = py::none());
    m.def("self_play_step_inplace", &self_play_step_inplace, py::arg("board"), py::arg("marks_black"), py::arg("marks_white"),
          py::arg("phase"), py::arg("current_player"), py::arg("pending_marks_required"), py::arg("pending_marks_remaining"),
          py::arg("pending_captures_required"), py::arg("pending_captures_remaining"), py::arg("forced_removals_done"),
          py::arg("move_count"), py::arg("moves_since_capture"), py::arg("plies"), py::arg("done"), py::arg("active_idx"),
          py::arg("chosen_action_codes"), py::arg("terminal_mask"), py::arg("chosen_valid_mask"), py::arg("max_game_plies"),
          py::arg("soft_value_k"));
    m.def("finalize_trajectory_inplace", &finalize_trajectory_inplace, py::arg("value_targets"), py::arg("soft_value_targets"),
          py::arg("player_signs"), py::arg("step_index_matrix"), py::arg("step_counts"), py::arg("slots"),
          py::arg("result_from_black"), py::arg("soft_value_from_black"));
    m.def("root_sparse_writeback", &root_sparse_writeback, py::arg("legal_index_mat"), py::arg("action_code_mat"),
          py::arg("valid_mask"), py::arg("legal_policy"), py::arg("local_picks"), py::arg("valid_root_indices"),
          py::arg("batch_size"), py::arg("total_action_dim"));
    m.def("postprocess_value_head", &postprocess_value_head, py::arg("raw_values"));
    m.def("apply_temperature_scaling", &apply_temperature_scaling, py::arg("probs"), py::arg("temperature"),
          py::arg("dim") = -1);
}
