"""Epoch loops of the RawGnn branch (reference ``Helpers/TrainTestHelper.py:12-159``).

Differences that do not change results: the running loss stays on the GPU and is read back once per epoch
(the reference syncs with ``loss.item()`` every step, ``TrainTestHelper.py:133``), and in a multi-process run
gradients are averaged across ranks by :class:`ihgnn_amd.distributed.GradientSync` before the optimiser step.
"""
import time
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.optim as optim

from ..Dataset import GraphDataset, TestSearchLogDataLoader
from .GlobalSettings import Gs
from .IOHelper import IOHelper
from .Metrics import Metrics
from .ProcessController import ProcessController


def print_network_parameters(module: nn.Module, name_filter: Optional[str] = None) -> None:
    rows = [(name, p) for name, p in module.named_parameters() if not name_filter or name_filter in name]
    if not rows:
        return
    shapes = ['(' + ', '.join(str(n) for n in p.size()) + ')' for _, p in rows]
    w_name, w_shape = max(len(n) for n, _ in rows), max(len(s) for s in shapes)
    with torch.no_grad():
        for (name, p), shape in zip(rows, shapes):
            grad = 'GRAD   ' if p.requires_grad else 'NO_GRAD'
            IOHelper.LogPrint(f'{name:<{w_name}} | size={shape:<{w_shape}} | {grad} | mean={p.mean().item():<7.3f} | '
                              f'std={p.std().item():<7.3f} | absmean={p.abs().mean().item():<7.3f}')


def _evaluate_batched(model, logs, item_count: int, device: torch.device, indices) -> List[Tuple[int, Metrics]]:
    """Top-10 of many searches per launch (``ihg_score_topk``: scores on the matrix cores, running top-10 in registers, no
    ``[C, I]`` matrix), one D2H copy per chunk (the reference scores one log at a time and syncs on every one,
    ``TrainTestHelper.py:58-67``, ``Metrics.py:60-61``)."""
    out: List[Tuple[int, Metrics]] = []
    chunk = 8192
    k = min(10, item_count)
    for lo in range(0, len(indices), chunk):
        part = indices[lo:lo + chunk]
        uq = torch.tensor([(logs[i][0], logs[i][1]) for i in part], dtype=torch.long, device=device)
        top = model.top_items(uq[:, 0], uq[:, 1], k)[0].tolist()          # HIP kernel: fp32-MFMA scores reduced to a running top-10 on chip
        for i, row in zip(part, top):
            _, _, items, flags, all_1 = logs[i]
            out.append((i, Metrics.from_top_indices(row, items, flags, all_1)))
    return out


def test_and_get_avg_metrics(model, dataset_train: GraphDataset, dataloader: TestSearchLogDataLoader,
                             get_long_tail_stat: bool = False) -> Tuple[Optional[List[Optional[Metrics]]], Metrics, float]:
    """-> (per-user average metrics or None, average over all usable logs, seconds).

    Logs are independent units: in a multi-process run each rank scores its contiguous share and the three metric sums
    and the count are all-reduced, so every rank returns the global average."""
    import torch.distributed as dist
    from .. import distributed as ihg_dist
    started = time.time()
    logs = dataloader.logs
    device = dataloader.device
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    mine = list(ihg_dist.shard_range(len(logs), rank, world))
    per_user: List[List[Metrics]] = [[] for _ in range(dataset_train.user_count)] if get_long_tail_stat else []
    total = Metrics()
    with torch.no_grad():
        model.save_features_for_test()
        try:
            if hasattr(model, 'top_items') and not Gs.Prediction.use_cosine_similarity:
                scored = _evaluate_batched(model, logs, dataset_train.item_count, device, mine)
            else:
                scored = []
                for i in mine:
                    u, q, items, flags, all_1 = logs[i]
                    users = torch.tensor([u], device=device).expand(dataset_train.item_count)
                    queries = torch.tensor([q], device=device).expand(dataset_train.item_count)
                    scored.append((i, Metrics.calculate_on_all_items(model(users, queries, None), items, flags, all_1)))
        finally:
            model.clear_saved_feature()
    for i, m in scored:
        total.add_to_self(m)
        if get_long_tail_stat:
            per_user[logs[i][0]].append(m)
    sums = ihg_dist.all_reduce_sums([total.HitRatio_at10, total.NDCG_at10, total.MAP_at10, float(len(scored))], device)
    counted = int(round(sums[3]))
    average = Metrics(sums[0], sums[1], sums[2]).divide_and_get_new(max(counted, 1))
    user_avgs = None
    if get_long_tail_stat:                                  # per-user sums and counts, added up over the ranks' shares of the logs
        table = torch.zeros(dataset_train.user_count, 4, dtype=torch.float64)
        for user, ms in enumerate(per_user):
            for m in ms:
                table[user] += torch.tensor([m.HitRatio_at10, m.NDCG_at10, m.MAP_at10, 1.0], dtype=torch.float64)
        if world > 1:
            table = table.to(device)
            dist.all_reduce(table, op=dist.ReduceOp.SUM)
            table = table.cpu()
        user_avgs = [None if row[3] < 0.5 else Metrics(row[0], row[1], row[2]).divide_and_get_new(int(round(row[3])))
                     for row in table.tolist()]
    seconds = time.time() - started
    IOHelper.LogPrint(f'evaluation done in {seconds:<.2f} s over {counted} usable search logs.')
    IOHelper.LogPrint(average.to_string(highlight=True), put_time_in_single_line=True)
    IOHelper.LogPrint()
    return user_avgs, average, seconds


# ``record_step='auto'``: an eager step faster than this is bound by the host's launch rate (config C1: 0.56 - 0.86 ms eager against 0.23 ms replayed; C2, 1.9 ms,
# gains nothing) - such a model is trained from a recorded step without being asked to
AUTO_RECORD_BELOW_MS = 1.5
_AUTO_WARM_STEPS, _AUTO_TIMED_STEPS = 3, 4


def _record_mode(record_step) -> str:
    if record_step in (True, 'on', '1', 'yes'):
        return 'on'
    if record_step in (False, 'off', '0', 'no'):
        return 'off'
    if record_step in (None, 'auto', ''):
        return 'auto'
    raise ValueError(f'record_step: on | off | auto, got {record_step!r}')


def train_and_get_avg_loss(model, optimizer: optim.Optimizer, loss_function: nn.Module, dataset_train: GraphDataset,
                           dataloader_train, pc: ProcessController, device: torch.device,
                           grad_sync=None, record_step='auto') -> Tuple[float, float]:
    """One epoch over ``dataloader_train`` -> (average loss, seconds) - the reference's loop (``Helpers/TrainTestHelper.py:123-143``).  ``record_step`` (single
    process, fused loss, the HIP Adam): batches of the common size run as replays of ONE recorded step (``ihgnn_amd.captured_step``: worth it where the step is
    bound by the host's launch rate, i.e. on small graphs); other batch sizes - the epoch's last batch - run eagerly.  ``'auto'`` (default): the first steps of the
    first epoch run eagerly, steps 4 - 7 are timed, and the step is recorded when they took less than ``AUTO_RECORD_BELOW_MS`` each; ``'on'`` / ``'off'`` decide
    without measuring.  A model the recording refuses (a parameter without gradient) trains eagerly, with one log line."""
    started = time.time()
    loss_sum = torch.zeros((), dtype=torch.float32, device=device)
    batches = positives = 0
    sampler = getattr(dataloader_train, 'batch_sampler', None)
    if hasattr(sampler, 'set_epoch'):
        sampler.set_epoch(pc.CurrentEpoch)                  # data parallel: a fresh shared permutation per epoch (ShardedBatchSampler)
    mode = _record_mode(record_step)
    from ..optim import Adam as _HipAdam
    can_record = mode != 'off' and grad_sync is None and device.type == 'cuda' and isinstance(optimizer, _HipAdam)
    decision = getattr(model, '_record_decision', None) if can_record else False       # None: undecided (auto, still measuring); True / False
    if can_record and mode == 'on' and decision is not True and not getattr(model, '_record_refused', False):
        # an explicit 'on' overrides what an earlier 'auto' epoch measured (False = "not launch-bound"); only a REFUSED recording (the capture raised) stays off
        decision = model._record_decision = True
    if decision and getattr(model, '_recorded_step', None) is not None and model._recorded_step.stale(full=True):
        model._recorded_step = None                          # an IHG_* switch or ops flag changed since the recording (checked once per epoch; the per-step check is the cheap one)
    eager_seen, timed_from = getattr(model, '_auto_eager_steps', 0), None      # (persist across epochs: an epoch may be shorter than the measurement)
    for p_u, p_q, p_i, p_f, n_u, n_q, n_i, n_f in dataloader_train:
        positives += len(p_u)
        users, queries, items = torch.cat([p_u, n_u]), torch.cat([p_q, n_q]), torch.cat([p_i, n_i])
        flags = torch.cat([p_f, n_f]).float()
        fused = getattr(model, 'supports_fused_loss', None) and model.supports_fused_loss(loss_function)
        if decision and fused:
            recorded = getattr(model, '_recorded_step', None)
            if recorded is None or recorded.optimizer is not optimizer or recorded.stale():     # (stale: a hyper-parameter baked into the recording changed)
                from ..captured_step import CapturedTrainingStep
                try:
                    recorded = model._recorded_step = CapturedTrainingStep(model, optimizer, int(users.shape[0]), warmup_batch=(users, queries, items, flags))
                except (ValueError, RuntimeError) as refused:
                    # ValueError: the recording's own refusal (a parameter without gradient, ...); RuntimeError: the stream capture failed (an op that synchronises,
                    # an allocation the capture cannot hold) - under 'auto' nobody asked for a recording, so either way the run goes on eagerly; under 'on' a
                    # capture failure is the caller's to see
                    if mode == 'on' and isinstance(refused, RuntimeError):
                        raise
                    IOHelper.LogPrint(f'training step not recorded ({type(refused).__name__}: {refused}); training eagerly')
                    recorded, decision = None, False
                    model._recorded_step, model._record_decision, model._record_refused = None, False, True
                    optimizer.zero_grad(set_to_none=True)
            if recorded is not None and recorded.batch_rows == int(users.shape[0]):
                loss_sum += recorded.step(users, queries, items, flags)
                batches += 1
                continue
        # auto: a few eager steps are timed (the step alone, device drained on both sides - not the loader's work between steps) once the allocator and the workspaces are warm
        time_this = decision is None and fused and _AUTO_WARM_STEPS <= eager_seen < _AUTO_WARM_STEPS + _AUTO_TIMED_STEPS
        if time_this:
            torch.cuda.synchronize(device)
            timed_from = time.perf_counter()
        if getattr(model, '_recorded_step', None) is not None:
            optimizer.zero_grad(set_to_none=True)           # the last replay's gradients (in the recording's pool) must not be accumulated onto
        cotangent = getattr(grad_sync, 'mode', None) == 'cotangent'
        if cotangent and not fused:
            raise RuntimeError('--grad_sync cotangent exchanges the fused batch tail\'s row gradients: it needs the HIP model and a plain BCEWithLogitsLoss (use bucketed / sharded)')
        if fused:
            loss = model.bce_loss(users, queries, items, flags, cotangent_sync=grad_sync) if cotangent else model.bce_loss(users, queries, items, flags)
        else:
            loss = loss_function(model(users, queries, items), flags)
        loss_sum += loss.detach()
        if loss.is_cuda:
            from .. import ops
            ops.backward(loss)                               # loss.backward() with a cached root gradient (no fill launch per step)
        else:
            loss.backward()
        if grad_sync is not None:
            grad_sync.average_gradients()
        optimizer.step()
        if grad_sync is not None:
            grad_sync.zero_grad()           # keeps the .grad views into the flat all-reduce buffer
        else:
            optimizer.zero_grad()
        batches += 1
        if decision is None and fused:
            if time_this:
                torch.cuda.synchronize(device)
                model._auto_timed_seconds = getattr(model, '_auto_timed_seconds', 0.0) + (time.perf_counter() - timed_from)
            eager_seen += 1
            model._auto_eager_steps = eager_seen
            if eager_seen == _AUTO_WARM_STEPS + _AUTO_TIMED_STEPS:
                step_ms = model._auto_timed_seconds * 1e3 / _AUTO_TIMED_STEPS
                decision = model._record_decision = bool(step_ms < AUTO_RECORD_BELOW_MS)
                model._auto_step_ms = step_ms
                IOHelper.LogPrint(f'eager training step {step_ms:.2f} ms: ' + ('launch-bound - from here on ONE recorded step (hipGraph) is replayed' if decision
                                                                              else 'GPU-bound - steps stay eager'))
    avg_loss = loss_sum.item() / max(batches, 1)
    if getattr(grad_sync, 'mode', None) == 'cotangent' and grad_sync.world_size > 1:
        # the cotangent exchange never moves a parameter or a dense gradient between the ranks: the replicas stay identical because every rank takes the same deterministic
        # step.  Once per epoch that is CHECKED (one broadcast of the parameters); a drift - a non-deterministic kernel, a rank that skipped a step - is reported and
        # healed from rank 0 instead of growing silently
        drift = grad_sync.check_replicas()
        if drift != 0.0:
            IOHelper.LogPrint(f'WARNING: replicas drifted apart under --grad_sync cotangent (largest parameter difference {drift:.3e}); re-broadcasting rank 0\'s parameters')
            grad_sync.broadcast_parameters(0)
    if grad_sync is not None and grad_sync.world_size > 1:  # every rank reports (and schedules its learning rate on) the global average
        from .. import distributed as ihg_dist
        total, count = ihg_dist.all_reduce_sums([avg_loss * batches, float(batches)], device)
        avg_loss = total / max(count, 1.0)
    seconds = time.time() - started
    IOHelper.LogPrint(f'[Epoch \033[0;44m{pc.CurrentEpoch:>2d}/{pc.EndEpoch - 1}\033[0m] average loss '
                      f'\033[0;45m{avg_loss:<.4f}\033[0m on {positives} interactions in {seconds:<.2f} s '
                      f'(remaining {pc.GetRemainingTimeString()}).')
    if Gs.adjust_learning_rate and avg_loss < 0.008 and Gs.learning_rate > 0.0004:
        Gs.learning_rate *= 0.98
        for group in optimizer.param_groups:
            group['lr'] = Gs.learning_rate
        IOHelper.LogPrint(f'learning rate -> {Gs.learning_rate}')
    return avg_loss, seconds
