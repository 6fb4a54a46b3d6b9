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


def test_and_get_avg_metrics(model, dataset_train: GraphDataset, dataloader: TestSearchLogDataLoader,
                             get_long_tail_stat: bool = False) -> Tuple[Optional[List[Optional[Metrics]]], Metrics, float]:
    """-> (per-user average metrics or None, average over all usable logs, seconds)."""
    started = time.time()
    total, counted = Metrics(), 0
    per_user: List[List[Metrics]] = [[] for _ in range(dataset_train.user_count)] if get_long_tail_stat else []
    with torch.no_grad():
        model.save_features_for_test()
        try:
            for users, queries, items_interacted, flags_interacted, flags_all_1 in dataloader:
                scores = model(users, queries, None)
                m = Metrics.calculate_on_all_items(scores, items_interacted, flags_interacted, flags_all_1)
                if m is None:
                    continue
                total.add_to_self(m)
                counted += 1
                if get_long_tail_stat:
                    per_user[int(users[0])].append(m)
        finally:
            model.clear_saved_feature()
    average = total.divide_and_get_new(max(counted, 1))
    user_avgs = None
    if get_long_tail_stat:
        user_avgs = []
        for ms in per_user:
            if not ms:
                user_avgs.append(None)
                continue
            acc = Metrics()
            for m in ms:
                acc.add_to_self(m)
            user_avgs.append(acc.divide_and_get_new(len(ms)))
    seconds = time.time() - started
    IOHelper.LogPrint(f'evaluation done in {seconds:<.2f} s over {counted} usable search logs.')
    IOHelper.LogPrint(average.to_string(highlight=True), put_time_in_single_line=True)
    IOHelper.LogPrint()
    return user_avgs, average, seconds


def train_and_get_avg_loss(model, optimizer: optim.Optimizer, loss_function: nn.Module, dataset_train: GraphDataset,
                           dataloader_train, pc: ProcessController, device: torch.device,
                           grad_sync=None) -> Tuple[float, float]:
    """One epoch over ``dataloader_train`` -> (average loss, seconds)."""
    started = time.time()
    loss_sum = torch.zeros((), dtype=torch.float32, device=device)
    batches = positives = 0
    for p_u, p_q, p_i, p_f, n_u, n_q, n_i, n_f in dataloader_train:
        positives += len(p_u)
        users, queries, items = torch.cat([p_u, n_u]), torch.cat([p_q, n_q]), torch.cat([p_i, n_i])
        flags = torch.cat([p_f, n_f]).float()
        loss = loss_function(model(users, queries, items), flags)
        loss_sum += loss.detach()
        loss.backward()
        if grad_sync is not None:
            grad_sync.average_gradients()
        optimizer.step()
        if grad_sync is not None:
            grad_sync.zero_grad()           # keeps the .grad views into the flat all-reduce buffer
        else:
            optimizer.zero_grad()
        batches += 1
    avg_loss = loss_sum.item() / max(batches, 1)
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
