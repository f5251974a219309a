"""Child process of tests/test_gpu_sharded.py::test_sharded_step_world2_on_one_card: one rank of a 2-rank job
(gloo rendezvous on 127.0.0.1, both ranks on the one MI355X of the box).  Each rank owns rows r % 2 == rank of the
feature table, builds ITS batches with the device sampler (seed 1 + rank), fetches their rows through the
device-resident sharded lookup (pc_shard_bucket + two all_to_all rounds + HIP row gather) and runs the fused step;
the same batch over the replicated table must give the same loss and gradients bit for bit (the gathered rows are the
same numbers; the step is deterministic).  Writes {"ok": bool, ...} as JSON to argv[3]."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, port, out = int(sys.argv[1]), sys.argv[2], sys.argv[3]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0",
                      PC_DIST_BACKEND="gloo", PC_FORCE_DEVICE="0")
    res = {"ok": False, "rank": rank}
    try:
        from types import SimpleNamespace
        import torch
        import torch.distributed as dist
        from p_companion_amd import distributed as pdist
        from p_companion_amd.data import SimilarityIndexLoader, generate_scaled_bpg
        from p_companion_amd.product2vec import Product2Vec
        r, w, _ = pdist.init_from_env("cuda")
        assert (r, w) == (rank, 2)
        dev = torch.device("cuda", 0)
        cfg = SimpleNamespace(PRODUCT_EMB_DIM=128, TYPE_EMB_DIM=64, HIDDEN_SIZE=256, NUM_ATTENTION_HEADS=4, DROPOUT=0.0,
                              MARGIN=1.0, DEVICE=dev)
        bpg = generate_scaled_bpg(20000, 100, seed=0)
        table = bpg.cuda(dev)["features"]
        sharded = pdist.ShardedFeatureTable(pdist.ShardedFeatureTable.shard(table, rank, 2), bpg.num_products, rank, 2)
        ld_s = SimilarityIndexLoader(bpg, 1024, seed=1 + rank, drop_last=True, device=dev, sharded=sharded)
        ld_r = SimilarityIndexLoader(bpg, 1024, seed=1 + rank, drop_last=True, device=dev)
        torch.manual_seed(0)
        m_s, m_r = Product2Vec(cfg).to(dev).train(), Product2Vec(cfg).to(dev).train()
        m_r.load_state_dict(m_s.state_dict())
        steps = 0
        worst = 0.0
        for bs, br in zip(ld_s, ld_r):
            assert bs["table"].shape == (2 * sharded.capacity, 128)
            ls = m_s.train_step_indexed(bs["table"], bs)
            lr = m_r.train_step_indexed(table, br)
            gs, gr = m_s.flatten_parameters()[1], m_r.flatten_parameters()[1]
            worst = max(worst, float((gs - gr).abs().max()), abs(float(ls) - float(lr)))
            pdist.all_reduce_mean_(gs, 2)                       # the data-parallel gradient exchange, both ranks
            steps += 1
            if steps == 3:
                break
        res.update(ok=bool(worst == 0.0 and sharded.overflowed() == 0), worst=worst, steps=steps,
                   capacity=sharded.capacity, bytes_per_peer=sharded.bytes_per_peer)
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:                                       # noqa: BLE001 -- reported to the parent test
        import traceback
        res["error"] = traceback.format_exc()
    with open(out, "w") as f:
        json.dump(res, f)


if __name__ == "__main__":
    main()
