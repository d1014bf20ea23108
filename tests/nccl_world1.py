"""Child process of tests/test_gpu_distributed.py::test_rccl_group_of_one...: what a 1-GPU box can execute of the RCCL
path -- library load, communicator creation, the device-tensor collectives of `gather_trajectories` (counts all-gather) and
`broadcast_checkpoint` (flat broadcast), bench.py's MAX all-reduce, and a grouped send / receive to itself -- in a
world-size-1 `nccl` group (backend "nccl" IS RCCL on ROCm).  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

FIELDS = ("state_tensors", "legal_masks", "policy_targets", "value_targets", "soft_value_targets")


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29611")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    out = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    try:
        from liuzhou_amd.distributed import broadcast_checkpoint, gather_trajectories
        from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
        from liuzhou_amd.self_play_gpu_runner import self_play_v1_gpu
        torch.manual_seed(7)
        model = ChessNet(**MODEL_CONFIGS["tiny"]).eval().to(dev)
        mine, _ = self_play_v1_gpu(model, num_games=5, mcts_simulations=8, temperature_init=1.0, temperature_final=0.1,
                                   temperature_threshold=10, exploration_weight=1.0, device="cuda:0",
                                   add_dirichlet_noise=True, sample_moves=True, concurrent_games=5, max_game_plies=12,
                                   autocast_dtype="float32")
        got = gather_trajectories(mine, dst=0, compact=True, force=True)          # counts all-gather on DEVICE tensors
        same = got is not None and got.num_samples == mine.num_samples and got.state_tensors.is_cuda
        for f in FIELDS:
            a, b = getattr(got, f), getattr(mine, f)
            same = same and (torch.equal(a, b) if a.dtype == torch.bool else torch.equal(a.view(torch.int32), b.view(torch.int32)))
        out["gather_bit_exact"] = bool(same)
        out["rows"] = int(mine.num_samples)
        big = ChessNet(**MODEL_CONFIGS["b6c64"]).to(dev)
        before = float(sum(p.double().sum() for p in big.parameters()) + sum(b.double().sum() for b in big.buffers()))
        broadcast_checkpoint(big, src=0, force=True)                              # flat DEVICE buffers through RCCL
        after = float(sum(p.double().sum() for p in big.parameters()) + sum(b.double().sum() for b in big.buffers()))
        out["broadcast_keeps_weights"] = before == after
        t = torch.tensor([3.25], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out["all_reduce_max"] = float(t.item())
        dist.barrier(device_ids=[0])
        out["barrier"] = True
        try:                                                                      # grouped ncclSend / ncclRecv to itself
            src = torch.arange(1024, dtype=torch.uint8, device=dev)
            dst = torch.zeros_like(src)
            for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, src, 0), dist.P2POp(dist.irecv, dst, 0)]):
                req.wait()
            torch.cuda.synchronize(dev)
            out["self_send_recv"] = bool(torch.equal(src, dst))
        except Exception as exc:  # informative only: not every RCCL build accepts a self-send
            out["self_send_recv"] = f"unsupported: {type(exc).__name__}"
    finally:
        dist.destroy_process_group()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
