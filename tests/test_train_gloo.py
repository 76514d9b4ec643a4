"""CPU, world_size=2 over gloo: the data-parallel logic of msae.train.SaeTrainStep -- asynchronous
per-parameter gradient all-reduce launched from post-accumulate-grad hooks, averaging, did_fire MAX
reduce, token counting, micro-batch chunking, gradient accumulation with per-batch clipping, the
warm-up / decay schedule -- is the product code under test.  The local compute (Sae.forward, the fused
clip + projection + Adam pass) runs on HIP only, so it is replaced here by torch-CPU restatements of the
same maths (test infrastructure, like the oracle in test_sharded_gloo.py).

Claim checked: after several steps every rank holds bit-identical parameters, and they equal (to fp32
summation order) a single process that minimises the AVERAGE of the ranks' losses -- DDP semantics
(train/sae/sae/trainer.py:338-345).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make(d=32, N=256, k=4):
    for p in (REPO, REPO / "tests", REPO / "multimodal-sae_amd"):
        if str(p) not in sys.path:
            sys.path.insert(0, str(p))
    import synth
    from msae import Sae, SaeConfig
    from msae.sae.sae import ForwardOutput
    from msae.train import SaeTrainStep

    W_enc, b_enc, W_dec, b_dec = synth.sae_weights(d, N, seed=51)
    sae = Sae(d, SaeConfig(num_latents=N, k=k), device="cpu")
    with torch.no_grad():
        sae.encoder.weight.copy_(torch.from_numpy(W_enc)); sae.encoder.bias.copy_(torch.from_numpy(b_enc))
        sae.W_dec.copy_(torch.from_numpy(W_dec)); sae.b_dec.copy_(torch.from_numpy(b_dec))

    def dense_forward(sae, x):
        """sae.py:193-247 without the AuxK / Multi-TopK terms, in dense torch ops."""
        pre = torch.relu((x - sae.b_dec) @ sae.encoder.weight.T + sae.encoder.bias)
        acts, idx = pre.topk(sae.cfg.k, dim=-1)
        out = (sae.W_dec[idx] * acts[..., None]).sum(1) + sae.b_dec
        fvu = (out - x).pow(2).sum() / (x - x.mean(0)).pow(2).sum()
        zero = out.new_tensor(0.0)
        return ForwardOutput(out, acts, idx, fvu, zero, zero)

    class CpuTrainStep(SaeTrainStep):
        halves = 1        # > 1: minimise the average loss of that many equal parts of every chunk

        def _forward(self, hiddens, dead_mask):
            parts = [dense_forward(self.sae, h) for h in hiddens.chunk(self.halves)]
            fvu = sum(p.fvu for p in parts) / len(parts)
            return ForwardOutput(parts[0].sae_out, torch.cat([p.latent_acts for p in parts]),
                                 torch.cat([p.latent_indices for p in parts]), fvu, parts[0].auxk_loss,
                                 parts[0].multi_topk_fvu)

        def _update(self, lr):
            torch.nn.utils.clip_grad_norm_(self.params, self.max_grad_norm)
            self.sae.remove_gradient_parallel_to_decoder_directions()
            b1, b2 = self.betas
            with torch.no_grad():
                for p, m, v in zip(self.params, self.exp_avg, self.exp_avg_sq):
                    m.mul_(b1).add_(p.grad, alpha=1 - b1)
                    v.mul_(b2).addcmul_(p.grad, p.grad, value=1 - b2)
                    denom = (v / (1 - b2 ** self.t)).sqrt() + self.eps
                    p.addcdiv_(m / (1 - b1 ** self.t), denom, value=-lr)
                    p.grad = None

    return sae, CpuTrainStep, synth


STEPS, T_RANK, D = 5, 24, 32


def _worker(rank, world, port, out_dir, grad_acc):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sae, CpuTrainStep, synth = _make()
    ts = CpuTrainStep(sae, lr=1e-2, micro_acc_steps=2, grad_acc_steps=grad_acc, lr_warmup_steps=2, total_steps=10,
                      init_b_dec=True)
    lrs, fvus = [], []
    for s in range(STEPS):
        x = torch.from_numpy(synth.activations(T_RANK, D, seed=100 + 2 * s + rank, bf16=False, n_outlier=1))
        lrs.append(ts.current_lr)
        fvus.append(float(ts.step(x)["fvu"]))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), lrs=np.array(lrs), fvus=np.array(fvus), t=ts.t,
             since=ts.num_tokens_since_fired.numpy(),
             **{n: p.detach().numpy() for n, p in sae.named_parameters()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("grad_acc", [1, 2])
def test_data_parallel_train_step_equals_average_loss_training(tmp_path, grad_acc):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), grad_acc), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    sae, CpuTrainStep, synth = _make()
    names = [n for n, _ in sae.named_parameters()]
    for n in names:
        assert np.array_equal(r0[n], r1[n]), f"{n}: ranks diverged"
    assert np.array_equal(r0["since"], r1["since"])
    # single process on the concatenated batches, minimising the average of the two ranks' losses
    ref = CpuTrainStep(sae, lr=1e-2, micro_acc_steps=2, grad_acc_steps=grad_acc, lr_warmup_steps=2, total_steps=10,
                       init_b_dec=True)
    ref.halves = 2
    lrs = []
    for s in range(STEPS):
        xs = [synth.activations(T_RANK, D, seed=100 + 2 * s + r, bf16=False, n_outlier=1) for r in range(world)]
        # chunk c of the concatenation must hold chunk c of BOTH ranks (each rank chunks its own batch in two)
        halves = [np.split(x, 2) for x in xs]
        x = torch.from_numpy(np.concatenate([halves[0][0], halves[1][0], halves[0][1], halves[1][1]]))
        lrs.append(ref.current_lr)
        ref.step(x)
    assert np.allclose(lrs, r0["lrs"]) and int(r0["t"]) == ref.t == STEPS // grad_acc
    assert lrs[0] == 0.0 and (grad_acc == 2 or lrs[2] == pytest.approx(1e-2))     # warm-up then decay
    for n, p in sae.named_parameters():
        np.testing.assert_allclose(r0[n], p.detach().numpy(), rtol=2e-5, atol=2e-6, err_msg=n)
    assert np.array_equal(r0["since"], ref.num_tokens_since_fired.numpy())
    assert (r0["since"] == 0).any() and (r0["since"] > 0).any()


def test_linear_schedule_matches_transformers():
    for p in (REPO / "multimodal-sae_amd",):
        sys.path.insert(0, str(p))
    from transformers import get_linear_schedule_with_warmup

    from msae.train import linear_schedule_with_warmup

    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    sch = get_linear_schedule_with_warmup(opt, 3, 11)
    for step in range(14):
        assert sch.get_last_lr()[0] == pytest.approx(linear_schedule_with_warmup(step, 3, 11))
        opt.step(); sch.step()
