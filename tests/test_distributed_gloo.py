"""Data-parallel plumbing on CPU with the gloo backend, world size 2 (the GPU path uses the same code over RCCL).

The HIP kernels cannot run here, so the model under the wrapper is the CPU oracle (test infrastructure); what is being
tested is ihgnn_amd.distributed: flat-buffer gradient views, one all-reduce, averaging, parameter broadcast, sharding.
Contract: N-rank result == 1-rank result on the union batch.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ihgnn_amd import distributed as ihg_dist
from ihgnn_amd import synth


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _make_model(seed):
    from oracle import ihgnn_ref as ref
    w = synth.draw(30, 10, 40, 20, 300, seed=5)
    g = ref.HyperGraph(w.triples, 30, 10, 40)
    m = ref.OracleRawGnn(g, torch.from_numpy(w.bag_words + 1), torch.from_numpy(w.bag_offsets), 20, 8, 'ihgnn', 2, 3)
    torch.manual_seed(seed)
    with torch.no_grad():
        for p in m.parameters():
            p.normal_(0, 0.3)
    return m


def _batch():
    rng = np.random.default_rng(9)
    u, q, i = (torch.from_numpy(rng.integers(0, n, 64)) for n in (30, 10, 40))
    y = torch.from_numpy((rng.random(64) < 0.3).astype(np.float32))
    return u, q, i, y


def _worker(rank, world, port, out_dir, mode='flat'):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    got_rank, _, got_world = ihg_dist.init_from_env('gloo')
    assert (got_rank, got_world) == (rank, world)
    torch.set_num_threads(1)
    model = _make_model(seed=100 + rank)                    # replicas start DIFFERENT ...
    sync = ihg_dist.make_gradient_sync(model, mode)
    sync.broadcast_parameters(0)                            # ... and are made identical to rank 0
    u, q, i, y = _batch()
    rows = ihg_dist.shard_range(64, rank, world)
    sl = slice(rows.start, rows.stop)
    opt = sync.optimizer(1e-2) if sync.owns_optimizer else torch.optim.Adam(model.parameters(), 1e-2)
    lossf = torch.nn.BCEWithLogitsLoss()
    for step in range(2):                                   # two steps: the .grad views must survive zero_grad
        loss = lossf(model(u[sl], q[sl], i[sl]), y[sl])
        loss.backward()
        assert all(p.grad.data_ptr() >= sync.flat.data_ptr() for p in sync.params)      # grads live in the flat buffer
        if mode == 'bucketed':
            assert all(sync._launched), 'every bucket was launched by its hook during the backward'
        sync.average_gradients()
        opt.step()
        sync.zero_grad()
        if mode == 'sharded' and step == 0:                 # the checkpoint of the sharded optimizer holds EVERY rank's Adam state
            state = opt.state_dict()
            # torch.optim.Adam's own layout (what the reference and the flat / bucketed modes write): a plain Adam over the same parameters loads it ...
            assert set(state) == {'state', 'param_groups'} and len(state['state']) == len(sync.params)
            assert sum(e['exp_avg'].numel() for e in state['state'].values()) == sync.flat.numel() and sum(float(e['exp_avg'].abs().sum()) for e in state['state'].values()) > 0
            plain = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in sync.params], 1e-2)
            plain.load_state_dict(state)
            # ... and the sharded optimizer loads a plain Adam's checkpoint (a run resumed under another --grad_sync mode) as well as its own
            fresh = sync.optimizer(1e-2)
            fresh.load_state_dict(plain.state_dict())
            lo, hi = sync.shard_range
            mine = next(iter(fresh.inner.state_dict()['state'].values()))
            want = next(iter(opt.inner.state_dict()['state'].values()))
            assert torch.equal(mine['exp_avg'], want['exp_avg']) and torch.equal(mine['exp_avg_sq'], want['exp_avg_sq'])
            opt = fresh
        if mode == 'sharded' and step == 1:
            # the driver's checkpoint pattern (Main.py): EVERY rank builds the state - state_dict() of the sharded optimizer is a collective, a
            # chief-only call would wait for its peers forever - and only the chief writes it; the file is complete: the other rank finds its shard in it
            state = ihg_dist.checkpoint_state(7, model, opt)
            path = os.path.join(out_dir, 'checkpoint_chief.pt')
            if rank == 0:
                torch.save(state, path)
            dist.barrier()
            saved = torch.load(path)
            assert saved['epoch_count'] == 7 and set(saved) == {'epoch_count', 'model', 'optimizer'}
            lo, hi = sync.shard_range
            mine = next(iter(opt.inner.state_dict()['state'].values()))
            hi_real = min(hi, sync.flat.numel())
            flat_saved = torch.cat([saved['optimizer']['state'][k]['exp_avg'].reshape(-1) for k in sorted(saved['optimizer']['state'])])
            assert torch.equal(flat_saved[lo:hi_real], mine['exp_avg'].reshape(-1)[:hi_real - lo])
            assert set(opt.local_state_dict()) == {'adam_shard', 'shard_range', 'numel'}           # the collective-free form
    torch.save({k: v.clone() for k, v in model.reference_state().items()}, os.path.join(out_dir, f'rank{rank}.pt'))
    if mode == 'bucketed':                                  # a second backward before the exchange was waited for must not pass silently
        lossf(model(u[sl], q[sl], i[sl]), y[sl]).backward()
        with pytest.raises(RuntimeError, match='already launched'):
            lossf(model(u[sl], q[sl], i[sl]), y[sl]).backward()
        sync.average_gradients()
    sums = ihg_dist.all_reduce_sums([float(rank + 1), 10.0], torch.device('cpu'))
    assert sums == [3.0, 20.0]
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('mode', ['flat', 'bucketed', 'sharded'])
def test_two_rank_training_equals_single_rank_on_union_batch(tmp_path, mode):
    """flat = one all-reduce; bucketed = per-bucket all-reduces launched from the backward hooks; sharded = reduce-scatter, Adam on the
    rank's shard, all-gather of the parameters.  All three: replicas stay identical and equal the 1-rank run on the union batch."""
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), mode), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f'rank{k}.pt') for k in (0, 1))
    for k in r0:
        assert torch.equal(r0[k], r1[k]), f'replicas diverged on {k}'
    # single process, union batch, same start (rank 0's weights)
    model = _make_model(seed=100)
    u, q, i, y = _batch()
    opt = torch.optim.Adam(model.parameters(), 1e-2)
    lossf = torch.nn.BCEWithLogitsLoss()
    for _ in range(2):
        lossf(model(u, q, i), y).backward()
        opt.step(); opt.zero_grad()
    for k, v in model.reference_state().items():
        np.testing.assert_allclose(r0[k].numpy(), v.numpy(), rtol=2e-5, atol=2e-6, err_msg=k)


def _cotangent_worker(rank, world, port, out_dir, unequal):
    """The cotangent exchange with the CPU oracle under it: the fused batch tail's row gradients are emulated with torch (per-row cotangents of the propagation's output +
    the items-bias column), ``CotangentSync`` gathers rows and row gradients, the one propagation backward runs on the union."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    ihg_dist.init_from_env('gloo')
    torch.set_num_threads(1)
    model = _make_model(seed=100 + rank)
    sync = ihg_dist.make_gradient_sync(model, 'cotangent')
    assert sync.mode == 'cotangent' and not sync.owns_optimizer and sync.world_size == world
    sync.broadcast_parameters(0)
    u, q, i, y = _batch()
    cut = 35 if unequal else 32                              # rank 0: rows [0, cut), rank 1: [cut, 64)
    sl = slice(0, cut) if rank == 0 else slice(cut, 64)
    g = model.g
    opt = torch.optim.Adam(model.parameters(), 1e-2)
    lossf = torch.nn.BCEWithLogitsLoss()
    bias = model.P('prediction_layer.items_bias')
    for step in range(2):
        rows = torch.cat([u[sl], q[sl] + g.user_count, i[sl] + g.user_count + g.query_count])
        union_rows = sync.gather_rows(rows)
        b = rows.shape[0] // 3
        feats = model.propagate()
        picked = feats.detach()[rows].requires_grad_(True)
        item_bias = bias.detach()[i[sl]].requires_grad_(True)
        from oracle import ihgnn_ref as ref
        loss = lossf(ref.hem_score(picked[:b], picked[b:2 * b], picked[2 * b:], item_bias, model.lam), y[sl])
        loss.backward()
        rowgrad = torch.zeros(3 * b, feats.shape[1] + 4)
        rowgrad[:, :feats.shape[1]] = picked.grad / world
        rowgrad[2 * b:, feats.shape[1]] = item_bias.grad / world
        got_rows, union = sync.exchange(rows, rowgrad)
        assert got_rows is union_rows and union.shape[0] == union_rows.shape[0] == 3 * world * max(cut, 64 - cut)
        assert sync.sent_bytes == (3 * max(cut, 64 - cut)) * (8 + 4 * rowgrad.shape[1]) and sync.exchanged_bytes == (world - 1) * sync.sent_bytes
        third = union_rows.shape[0] // 3
        assert int(union_rows[:third].max()) < g.user_count <= int(union_rows[third:2 * third].min()) and int(union_rows[2 * third:].min()) >= g.user_count + g.query_count
        cot = torch.zeros_like(feats).index_add_(0, union_rows, union[:, :feats.shape[1]])
        feats.backward(cot)
        bias.grad = torch.zeros_like(bias).index_add_(0, union_rows[2 * third:] - g.user_count - g.query_count, union[2 * third:, feats.shape[1]])
        sync.average_gradients()                             # (nothing left to do)
        opt.step()
        sync.zero_grad()
        assert all(p.grad is None for p in model.parameters())
    assert sync.check_replicas() == 0.0                      # bitwise identical replicas without any parameter / gradient exchange
    torch.save({k: v.clone() for k, v in model.reference_state().items()}, os.path.join(out_dir, f'rank{rank}.pt'))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('unequal', [False, True])
def test_two_rank_cotangent_exchange_equals_single_rank_on_union_batch(tmp_path, unequal):
    """``--grad_sync cotangent``: the ranks all-gather the batch rows' cotangents (3B (D + 1) floats each) instead of all-reducing the dense gradients, and each runs the
    propagation backward on the union: replicas stay bitwise identical and equal the 1-rank run on the mean of the ranks' losses (``Models/RawGnn.py:122-142``: the batch
    reads F at 3B rows; ``Helpers/TrainTestHelper.py:126-143``: the loss is a mean over the batch).  ``unequal``: batches of 35 and 29 rows - the shorter is padded with
    zero-cotangent rows inside the exchange."""
    port = _free_port()
    mp.spawn(_cotangent_worker, args=(2, port, str(tmp_path), unequal), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f'rank{k}.pt') for k in (0, 1))
    for k in r0:
        assert torch.equal(r0[k], r1[k]), f'replicas diverged on {k}'
    model = _make_model(seed=100)
    u, q, i, y = _batch()
    cut = 35 if unequal else 32
    opt = torch.optim.Adam(model.parameters(), 1e-2)
    lossf = torch.nn.BCEWithLogitsLoss()
    for _ in range(2):
        (0.5 * (lossf(model(u[:cut], q[:cut], i[:cut]), y[:cut]) + lossf(model(u[cut:], q[cut:], i[cut:]), y[cut:]))).backward()
        opt.step(); opt.zero_grad()
    for k, v in model.reference_state().items():
        np.testing.assert_allclose(r0[k].numpy(), v.numpy(), rtol=2e-5, atol=2e-6, err_msg=k)


def test_cotangent_sync_single_process_is_transparent():
    """One rank, no process group: the exchange hands back what it was given (rows re-laid as thirds = unchanged), nothing is sent."""
    lin = torch.nn.Linear(4, 3)
    sync = ihg_dist.CotangentSync(lin.parameters())
    rows = torch.tensor([0, 1, 5, 6, 9, 9])
    union = sync.gather_rows(rows)
    assert torch.equal(union, rows) and union.as_int32.dtype == torch.int32
    rg = torch.randn(6, 7)
    got_rows, got = sync.exchange(rows, rg)
    assert got_rows is union and torch.equal(got, rg) and sync.exchanged_bytes == 0
    with pytest.raises(RuntimeError, match='gather_rows'):
        sync.exchange(rows, rg)
    assert ihg_dist.choose_gradient_sync(10 << 30, 8, True) == 'cotangent' and ihg_dist.choose_gradient_sync(10 << 30, 8, False) == 'sharded'
    assert ihg_dist.choose_gradient_sync(200 << 20, 8, False) == 'bucketed'
    c3 = ihg_dist.cotangent_bytes_per_rank(1100, 512)
    assert c3 == 3300 * (8 + 4 * 516) and ihg_dist.choose_gradient_sync(194_600_000, 8, True, c3) == 'cotangent'      # C3: 6.8 MB a rank against a 194.6 MB gradient
    assert ihg_dist.choose_gradient_sync(725_152, 2, True, ihg_dist.cotangent_bytes_per_rank(1100, 128)) == 'bucketed'  # C1: the whole gradient is smaller than the cotangents


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 64, 1001):
        for world in (1, 2, 3, 8):
            parts = [ihg_dist.shard_range(n, r, world) for r in range(world)]
            assert [x for p in parts for x in p] == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_gradient_sync_single_process_is_transparent():
    lin = torch.nn.Linear(4, 3)
    sync = ihg_dist.GradientSync(lin.parameters())
    lin(torch.ones(2, 4)).sum().backward()
    want = [p.grad.clone() for p in lin.parameters()]
    sync.average_gradients()
    assert all(torch.equal(p.grad, w) for p, w in zip(lin.parameters(), want))
    lin.zero_grad()                                          # set_to_none detaches the views ...
    lin(torch.ones(2, 4)).sum().backward()
    sync.average_gradients()                                 # ... and they are re-attached here
    assert all(torch.equal(p.grad, w) for p, w in zip(lin.parameters(), want))
    assert lin.weight.grad.data_ptr() == sync.flat.data_ptr()


def test_bucketed_sync_layout_puts_dense_parameters_first():
    model = _make_model(seed=1)
    sync = ihg_dist.BucketedGradientSync(model.named_parameters())
    sizes = [e - b for b, e in sync.buckets]
    tables = [p.numel() for n, p in model.named_parameters() if 'embedding' in n]
    assert sizes[1:] == tables and sizes[0] == sum(p.numel() for n, p in model.named_parameters() if 'embedding' not in n)
    assert sum(sizes) == sync.flat.numel()


def test_sharded_batch_sampler_covers_each_epoch_exactly_once():
    for n, batch, world in ((1000, 100, 1), (1001, 100, 2), (1234, 100, 8), (99, 100, 2), (807, 50, 3)):
        for epoch in (0, 1):
            per_rank = []
            for rank in range(world):
                s = ihg_dist.ShardedBatchSampler(n, batch, rank, world, shuffle=True, seed=3)
                s.set_epoch(epoch)
                batches = list(s)
                assert len(batches) == len(s) and all(0 < len(b) <= batch for b in batches)
                per_rank.append(batches)
            assert len({len(b) for b in per_rank}) == 1                   # same number of steps (= all-reduces) on every rank
            seen = sorted(x for batches in per_rank for b in batches for x in b)
            assert seen == list(range(n))                                # every training row exactly once per epoch
        first = [list(ihg_dist.ShardedBatchSampler(n, batch, 0, world, seed=3))]
        s2 = ihg_dist.ShardedBatchSampler(n, batch, 0, world, seed=3); s2.set_epoch(1)
        assert n < 3 or first[0] != list(s2)                             # a fresh permutation per epoch


def test_sharded_checkpoint_is_numbered_like_adam_over_all_parameters():
    """The sharded optimizer's checkpoint uses ``torch.optim.Adam(model.parameters())``'s numbering (reference: written at Main.py:255-259, read at 208-212): a FROZEN
    parameter keeps its index and has no entry, a parameter that never received a gradient has no entry in a plain Adam's checkpoint and loads as zero moments."""
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(3, 4)), torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.randn(2, 2)), torch.nn.Parameter(torch.randn(6))]
    ps[1].requires_grad_(False)                              # frozen: index 1 stays in the numbering, is not exchanged
    sync = ihg_dist.ShardedGradientSync(ps)
    assert sync.param_index == [0, 2, 3] and sync.n_all_params == 4
    opt = sync.optimizer(1e-2)
    (ps[0].sum() * 2 + ps[2].pow(2).sum() + ps[3].sum()).backward()
    sync.average_gradients(); opt.step(); sync.zero_grad()
    state = opt.state_dict()
    assert sorted(state['state']) == [0, 2, 3] and state['param_groups'][0]['params'] == [0, 1, 2, 3]
    # a plain Adam over ALL parameters (what the reference / the flat mode builds) loads it, entry by entry on the right parameter
    clones = [torch.nn.Parameter(p.detach().clone(), requires_grad=p.requires_grad) for p in ps]
    plain = torch.optim.Adam(clones, 1e-2)
    plain.load_state_dict(state)
    assert tuple(plain.state[clones[2]]['exp_avg'].shape) == (2, 2) and clones[1] not in plain.state
    # a plain Adam's checkpoint in which parameter 3 never received a gradient (no entry): the sharded optimizer takes zeros for it
    plain2 = torch.optim.Adam(clones, 1e-2)
    (clones[0].sum() + clones[2].sum()).backward()
    plain2.step()
    partial = plain2.state_dict()
    assert sorted(partial['state']) == [0, 2]
    fresh = sync.optimizer(1e-2)
    fresh.load_state_dict(partial)
    got = fresh._gather_full()
    n0, n2 = ps[0].numel(), ps[2].numel()
    assert torch.equal(got['exp_avg'][:n0].view(3, 4), partial['state'][0]['exp_avg']) and torch.equal(got['exp_avg'][n0:n0 + n2].view(2, 2), partial['state'][2]['exp_avg'])
    assert float(got['exp_avg'][n0 + n2:].abs().sum()) == 0.0
    # state for a parameter this model does not train is refused, not paired with a neighbour
    bad = {'state': {1: partial['state'][0]}, 'param_groups': partial['param_groups']}
    with pytest.raises(ValueError, match='does not train'):
        sync.optimizer(1e-2).load_state_dict(bad)
