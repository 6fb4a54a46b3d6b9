"""Pin the CPU oracle (oracle/ihgnn_ref.py) to fixtures produced by the real reference.

Fixtures: tests/golden/*.npz|json, written by tests/golden/make_golden.py (imports /root/reference).
Tolerances: the oracle replays the reference's own ATen op sequence, so agreement is expected at
float32 round-off (<= 2e-6 relative); anything looser would not pin it.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ihgnn_ref as ref
from conftest import GOLDEN

RTOL = 2e-6


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def f1_graph():
    z = np.load(os.path.join(GOLDEN, 'f1_graph.npz'))
    U, Q, I, V, N = z['counts']
    return z, ref.HyperGraph(z['pos_uqif'][:, :3], int(U), int(Q), int(I))


def test_f1_graph_tensors():
    z, g = f1_graph()
    assert g.edge_count == int(z['EdgeCount'])
    np.testing.assert_array_equal(g.I3.numpy(), z['I3'])
    np.testing.assert_array_equal(g.Adjacency.indices().numpy(), z['coo_indices'])
    np.testing.assert_array_equal(g.Adjacency.values().numpy(), z['coo_values'])
    np.testing.assert_array_equal(g.VertexDegrees.numpy(), z['VertexDegrees'])
    np.testing.assert_array_equal(g.EdgeDegrees.numpy(), z['EdgeDegrees'])
    # the fixture really does contain an isolated node (degree replaced by 1e-8) and a duplicate edge
    assert (z['VertexDegrees'] == np.float32(1e-8)).sum() >= 2
    assert len({tuple(r) for r in z['I3'].tolist()}) < len(z['I3'])


def small_graph():
    w = np.load(os.path.join(GOLDEN, 'f2_small_workload.npz'))
    U, Q, I, V = (int(x) for x in w['counts'])
    return w, ref.HyperGraph(w['triples'], U, Q, I)


LAYER_CASES = [(tag, kind, order) for tag in ('tiny_d8', 'small_d64')
               for kind, order in (('ihgnn', 1), ('ihgnn', 2), ('ihgnn', 3), ('hgcn', 0))]


@pytest.mark.parametrize('tag,kind,order', LAYER_CASES)
def test_f2_layer_forward_backward(tag, kind, order):
    z = np.load(os.path.join(GOLDEN, 'f2_layers.npz'))
    g = f1_graph()[1] if tag == 'tiny_d8' else small_graph()[1]
    pre = f'{tag}.{kind}{order}.'
    t = lambda k: torch.from_numpy(z[pre + k]).clone().requires_grad_(True)
    x = t('x')
    wt, bt = t('sd.feature_transform.weight'), t('sd.feature_transform.bias')
    if kind == 'ihgnn':
        wa, ba = t('sd.feature_interactor.aggregation.weight'), t('sd.feature_interactor.aggregation.bias')
        y = ref.ihgnn_layer(x, g, wt, bt, wa, ba, order)
        params = {'feature_transform.weight': wt, 'feature_transform.bias': bt,
                  'feature_interactor.aggregation.weight': wa, 'feature_interactor.aggregation.bias': ba}
    else:
        y = ref.hgcn_layer(x, g, wt, bt)
        params = {'feature_transform.weight': wt, 'feature_transform.bias': bt}
    y.backward(torch.from_numpy(z[pre + 'cot']))
    assert rel_err(y.detach().numpy(), z[pre + 'y']) <= RTOL
    assert rel_err(x.grad.numpy(), z[pre + 'dx']) <= RTOL
    for name, p in params.items():
        assert rel_err(p.grad.numpy(), z[pre + 'grad.' + name]) <= RTOL, name


def build_model(z, tag, g, w, kind):
    L, order, d = (int(v) for v in z[f'{tag}.cfg'])
    m = ref.OracleRawGnn(g, torch.from_numpy(w['bag_words'] + 1), torch.from_numpy(w['bag_offsets']),
                         int(w['counts'][3]), d, kind, L, order)
    return m


@pytest.mark.parametrize('tag,kind', [('ihgnn', 'ihgnn'), ('hgcn', 'hgcn'), ('ihgnn_o2', 'ihgnn')])
def test_f3_model_scores_grads_adam(tag, kind):
    z = np.load(os.path.join(GOLDEN, 'f3_model.npz'))
    w, g = small_graph()
    m = build_model(z, tag, g, w, kind)
    m.load_reference_state({k[len(tag) + 4:]: z[k] for k in z.files if k.startswith(f'{tag}.sd.')})
    u, q, i = (torch.from_numpy(z[f'{tag}.{k}']) for k in 'uqi')
    opt = torch.optim.Adam(m.parameters(), 1e-3, weight_decay=0)
    scores = m(u, q, i)
    loss = torch.nn.BCEWithLogitsLoss()(scores, torch.from_numpy(z[f'{tag}.flags']))
    loss.backward()
    assert rel_err(scores.detach().numpy(), z[f'{tag}.scores']) <= RTOL
    assert abs(loss.item() - float(z[f'{tag}.loss'])) <= 1e-6
    for key, gr in m.reference_grads().items():
        assert rel_err(gr.numpy(), z[f'{tag}.grad.{key}']) <= 5e-6, key
    opt.step()
    for key, p in m.reference_state().items():
        assert rel_err(p.numpy(), z[f'{tag}.after.{key}']) <= 5e-6, key


@pytest.mark.parametrize('tag,kind', [('ihgnn', 'ihgnn'), ('hgcn', 'hgcn'), ('ihgnn_o2', 'ihgnn')])
def test_f3_eval_path(tag, kind):
    z = np.load(os.path.join(GOLDEN, 'f3_model.npz'))
    w, g = small_graph()
    m = build_model(z, tag, g, w, kind)
    m.load_reference_state({k[len(tag) + 4:]: z[k] for k in z.files if k.startswith(f'{tag}.sd.')})
    with torch.no_grad():
        m.save_features_for_test()
        assert rel_err(m._saved.numpy(), z[f'{tag}.features']) <= RTOL
        ones = torch.ones(g.item_count, dtype=torch.long)
        for (uu, qq), want in zip(z[f'{tag}.eval_uq'], z[f'{tag}.eval_scores']):
            got = m(int(uu) * ones, int(qq) * ones, None)
            assert rel_err(got.numpy(), want) <= RTOL
        m.clear_saved_feature()


@pytest.mark.parametrize('tag', ['d128_o3', 'd128_o2', 'd256_o3'])
def test_f8_wide_models(tag):
    """BASELINE configs[2..4] shapes (d = 128 x 3 layers, d = 256 x 2 layers) on the small graph, from the reference."""
    from conftest import f8_case, f8_gradient_error
    (L, order, d), sd, z = f8_case(tag)
    w, g = small_graph()
    m = ref.OracleRawGnn(g, torch.from_numpy(w['bag_words'] + 1), torch.from_numpy(w['bag_offsets']), int(w['counts'][3]), d, 'ihgnn', L, order)
    m.load_reference_state(sd)
    u, q, i = (torch.from_numpy(z[f'{tag}.{k}']) for k in 'uqi')
    scores = m(u, q, i)
    loss = torch.nn.BCEWithLogitsLoss()(scores, torch.from_numpy(z[f'{tag}.flags']))
    loss.backward()
    assert rel_err(scores.detach().numpy(), z[f'{tag}.scores']) <= RTOL
    assert abs(loss.item() - float(z[f'{tag}.loss'])) <= 1e-6
    for key, gr in m.reference_grads().items():
        assert f8_gradient_error(z, tag, key, gr.numpy()) <= 5e-6, key
    with torch.no_grad():
        assert rel_err(m.propagate().numpy(), z[f'{tag}.features']) <= RTOL


def f9_logs():
    """The rows of tests/golden/f9_data/train_data.csv as (user, query, items, flags) - parsed here in a few lines of Python so that
    the oracle test depends on nothing under ihgnn_amd."""
    rows = []
    with open(os.path.join(GOLDEN, 'f9_data', 'train_data.csv')) as f:
        next(f)
        for line in f:
            c = line.rstrip('\n').split(',')
            rows.append((int(c[0]), int(c[1]), [int(x) for x in c[3].split()], [int(x) for x in c[6].split()]))
    return rows


def test_f9_log_hypergraph_and_hgcn():
    """Variable-arity hyperedges (one per search log): graph tensors, HGCN layer forward / backward, RawGnn over it."""
    z = np.load(os.path.join(GOLDEN, 'f9_log_hypergraph.npz'))
    U, Q, I, V = (int(x) for x in z['counts'])
    g = ref.LogHyperGraph(f9_logs(), U, Q, I)
    assert g.edge_count == int(z['EdgeCount'])
    np.testing.assert_array_equal(g.Adjacency.indices().numpy(), z['adj_indices'])
    np.testing.assert_array_equal(g.Adjacency.values().numpy(), z['adj_values'])
    np.testing.assert_array_equal(g.VertexDegrees.numpy(), z['VertexDegrees'])
    np.testing.assert_array_equal(g.EdgeDegrees.numpy(), z['EdgeDegrees'])
    assert z['adj_values'].max() == 2.0                      # the repeated item of one log
    for d in (16, 64):
        x = torch.from_numpy(z[f'd{d}.x']).requires_grad_(True)
        w = torch.from_numpy(z[f'd{d}.sd.feature_transform.weight']).requires_grad_(True)
        b = torch.from_numpy(z[f'd{d}.sd.feature_transform.bias']).requires_grad_(True)
        y = ref.hgcn_layer(x, g, w, b)
        y.backward(torch.from_numpy(z[f'd{d}.cot']))
        assert rel_err(y.detach().numpy(), z[f'd{d}.y']) <= RTOL and rel_err(x.grad.numpy(), z[f'd{d}.dx']) <= RTOL
        assert rel_err(w.grad.numpy(), z[f'd{d}.grad.feature_transform.weight']) <= RTOL
        assert rel_err(b.grad.numpy(), z[f'd{d}.grad.feature_transform.bias']) <= RTOL
    m = ref.OracleRawGnn(g, torch.from_numpy(z['bag_input']), torch.from_numpy(z['bag_offsets']), V, 16, 'hgcn', 2, 1)
    m.load_reference_state({k[len('model.sd.'):]: z[k] for k in z.files if k.startswith('model.sd.')})
    scores = m(*(torch.from_numpy(z[f'model.{k}']) for k in 'uqi'))
    loss = torch.nn.BCEWithLogitsLoss()(scores, torch.from_numpy(z['model.flags']))
    loss.backward()
    assert rel_err(scores.detach().numpy(), z['model.scores']) <= RTOL and abs(loss.item() - float(z['model.loss'])) <= 1e-6
    for key, gr in m.reference_grads().items():
        assert rel_err(gr.numpy(), z[f'model.grad.{key}']) <= 5e-6, key


def test_f4_metrics_known_answers():
    rec = json.load(open(os.path.join(GOLDEN, 'f4_metrics.json')))
    sc = rec['selfcheck']
    hr, ndcg, ap = ref.ranking_metrics(torch.tensor(sc['scores']), sc['truth'])
    # the reference's own __main__ self-check (Helpers/Metrics.py:165-193): 1.0000 / 0.6653 / 0.5000
    assert (round(hr, 4), round(ndcg, 4), round(ap, 4)) == (1.0, 0.6653, 0.5)
    assert hr == sc['hr'] and abs(ndcg - sc['ndcg']) < 1e-12 and abs(ap - sc['map']) < 1e-12
    assert abs(sc['idcg3'] - 2.1309297535714573) < 1e-12 and abs(sc['idcg_211'] - 4.130929753571458) < 1e-12
    for c in rec['random_cases']:
        hr, ndcg, ap = ref.ranking_metrics(torch.tensor(c['scores']), c['truth'])
        assert abs(hr - c['hr']) < 1e-12 and abs(ndcg - c['ndcg']) < 1e-12 and abs(ap - c['map']) < 1e-12
    c = rec['graded']
    hr, ndcg, ap = ref.ranking_metrics(torch.tensor(c['scores']), c['truth'], c['flags'])
    assert abs(hr - c['hr']) < 1e-12 and abs(ndcg - c['ndcg']) < 1e-12 and abs(ap - c['map']) < 1e-12


def test_f4_epoch_schedule():
    rec = json.load(open(os.path.join(GOLDEN, 'f4_metrics.json')))
    got = ref.epoch_schedule(20, 5, 7, 2)
    assert [(r['epoch'], r['test'], r['store']) for r in rec['schedule_20_5_7_2']] == got
    # reference __main__ (Helpers/ProcessController.py:114-118): tests after 11,13,...,23,24
    assert [e for e, t, _ in got if t] == [11, 13, 15, 17, 19, 21, 23, 24]
    got = ref.epoch_schedule(12, 1, 3, 3, 12, 1000000)
    assert [(r['epoch'], r['test'], r['store']) for r in rec['schedule_12_1_3_3_store']] == got


@pytest.mark.parametrize('tag,kind,order', [('ihgnn3', 'ihgnn', 3), ('ihgnn1', 'ihgnn', 1), ('hgcn', 'hgcn', 1)])
def test_f5_config_c1(tag, kind, order):
    from ihgnn_amd import synth
    z = np.load(os.path.join(GOLDEN, 'f5_c1.npz'))
    w = synth.draw_config('C1')
    g = ref.HyperGraph(w.triples, w.user_count, w.query_count, w.item_count)
    m = ref.OracleRawGnn(g, torch.from_numpy(w.bag_words + 1), torch.from_numpy(w.bag_offsets), w.vocab_size, 64, kind, 1, order)
    sd = {k[len('ihgnn3.sd.'):]: z[k] for k in z.files if k.startswith('ihgnn3.sd.')}
    sd.update({k[len(tag) + 4:]: z[k] for k in z.files if k.startswith(f'{tag}.sd.')})
    m.load_reference_state(sd)
    u, q, i = (torch.from_numpy(z[f'{tag}.{k}']) for k in 'uqi')
    with torch.no_grad():
        assert rel_err(m(u, q, i).numpy(), z[f'{tag}.scores']) <= RTOL
        feats = m.propagate()
        assert rel_err(feats[torch.from_numpy(z[f'{tag}.rows'])].numpy(), z[f'{tag}.feat_rows']) <= RTOL
        assert rel_err(feats.double().sum(0).numpy(), z[f'{tag}.feat_colsum']) <= 1e-5


@pytest.mark.parametrize('tag', ['ihgnn', 'hgcn'])
def test_f6_training_curve_and_metrics(tag):
    z = np.load(os.path.join(GOLDEN, 'f6_training.npz'))
    w = np.load(os.path.join(GOLDEN, 'f6_workload.npz'))
    U, Q, I, V = (int(x) for x in w['counts'])
    g = ref.HyperGraph(w['triples'], U, Q, I)
    L, order, d = (int(v) for v in z[f'{tag}.cfg'])
    m = ref.OracleRawGnn(g, torch.from_numpy(w['bag_words'] + 1), torch.from_numpy(w['bag_offsets']), V, d, tag, L, order)
    m.load_reference_state({k[len(tag) + 6:]: z[k] for k in z.files if k.startswith(f'{tag}.init.')})
    opt = torch.optim.Adam(m.parameters(), 1e-3, weight_decay=0)
    lossf = torch.nn.BCEWithLogitsLoss()
    losses = []
    for b in z[f'{tag}.batches']:
        u, q, i, fl = (torch.from_numpy(b[k].astype(np.int64)) for k in range(4))
        loss = lossf(m(u, q, i), fl.float())
        loss.backward(); opt.step(); opt.zero_grad()
        losses.append(loss.item())
    np.testing.assert_allclose(losses, z[f'{tag}.losses'], rtol=2e-5, atol=0)
    ends = np.cumsum(w['test_items_len'])
    acc = np.zeros(3)
    with torch.no_grad():
        m.save_features_for_test()
        ones = torch.ones(I, dtype=torch.long)
        for k, (uu, qq) in enumerate(w['test_uq']):
            items = w['test_items_flat'][ends[k] - w['test_items_len'][k]:ends[k]].tolist()
            acc += ref.ranking_metrics(m(int(uu) * ones, int(qq) * ones, None), items)
    # north_star acceptance: HR@10 / NDCG@10 within +-0.002 of the reference
    np.testing.assert_allclose(acc / len(w['test_uq']), z[f'{tag}.metrics'], atol=2e-3)


@pytest.mark.parametrize('tag', ['d128_l3_o3', 'd64_l2_o3', 'd32_l2_o3', 'd128_l3_o2', 'd256_l2_o3'])
def test_f10_training_curve_and_metrics_at_the_headline_widths(tag):
    """F10: the reference's 48-step curve and ranking metrics at d = 128 x 3 layers (orders 3, 2), d = 64 x 2 and d = 32 x 2 (the reference's default
    width) on a power-law graph with split rows - the oracle replays it."""
    from conftest import f10_case, state_digest
    (L, order, d), sd, z, w = f10_case(tag)
    U, Q, I, V = (int(x) for x in w['counts'])
    g = ref.HyperGraph(w['triples'].astype(np.int64), U, Q, I)
    m = ref.OracleRawGnn(g, torch.from_numpy(w['bag_words'] + 1), torch.from_numpy(w['bag_offsets']), V, d, 'ihgnn', L, order)
    m.load_reference_state(sd)
    opt = torch.optim.Adam(m.parameters(), 1e-3, weight_decay=0)
    lossf = torch.nn.BCEWithLogitsLoss()
    losses = []
    for b in z[f'{tag}.batches']:
        u, q, i, fl = (torch.from_numpy(b[k].astype(np.int64)) for k in range(4))
        loss = lossf(m(u, q, i), fl.float())
        loss.backward(); opt.step(); opt.zero_grad()
        losses.append(loss.item())
    np.testing.assert_allclose(losses, z[f'{tag}.losses'], rtol=2e-5, atol=0)
    ends = np.cumsum(w['test_items_len'])
    acc = np.zeros(3)
    with torch.no_grad():
        m.save_features_for_test()
        ones = torch.ones(I, dtype=torch.long)
        for k, (uu, qq) in enumerate(w['test_uq']):
            items = w['test_items_flat'][ends[k] - w['test_items_len'][k]:ends[k]].tolist()
            acc += ref.ranking_metrics(m(int(uu) * ones, int(qq) * ones, None), items)
    np.testing.assert_allclose(acc / len(w['test_uq']), z[f'{tag}.metrics'], atol=2e-3)
    assert z[f'{tag}.metrics'][0] > 0.05                       # a trained model: the metrics are not the all-zero kind that any scores match


F7_CASES = [('tiny_uqi', 8, 'uqi'), ('small_uqi', 64, 'uqi'), ('small_ui', 32, 'ui'), ('tiny_qi', 8, 'qi')]


def pair_graph(tag, mode):
    if tag.startswith('tiny'):
        z, _ = f1_graph()
        U, Q, I, V, N = (int(x) for x in z['counts'])
        return ref.PairGraph(z['pos_uqif'][:, :3], U, Q, I, mode)
    w = np.load(os.path.join(GOLDEN, 'f2_small_workload.npz'))
    U, Q, I, V = (int(x) for x in w['counts'])
    return ref.PairGraph(w['triples'], U, Q, I, mode)


@pytest.mark.parametrize('tag,d,mode', F7_CASES)
def test_f7_gcn_layer(tag, d, mode):
    z = np.load(os.path.join(GOLDEN, 'f7_gcn.npz'))
    g = pair_graph(tag, mode)
    np.testing.assert_array_equal(g.Adjacency.indices().numpy(), z[f'{tag}.adj_indices'])
    np.testing.assert_array_equal(g.Adjacency.values().numpy(), z[f'{tag}.adj_values'])
    np.testing.assert_array_equal(g.VertexDegrees.numpy(), z[f'{tag}.degrees'])
    t = lambda k: torch.from_numpy(z[f'{tag}.{k}']).clone().requires_grad_(True)
    x, w, b = t('x'), t('sd.feature_transform.weight'), t('sd.feature_transform.bias')
    y = ref.gcn_layer(x, g, w, b)
    y.backward(torch.from_numpy(z[f'{tag}.cot']))
    assert rel_err(y.detach().numpy(), z[f'{tag}.y']) <= RTOL and rel_err(x.grad.numpy(), z[f'{tag}.dx']) <= RTOL
    assert rel_err(w.grad.numpy(), z[f'{tag}.grad.feature_transform.weight']) <= RTOL
    assert rel_err(b.grad.numpy(), z[f'{tag}.grad.feature_transform.bias']) <= RTOL


def test_f7_gcn_model():
    z = np.load(os.path.join(GOLDEN, 'f7_gcn.npz'))
    w, g = small_graph()
    m = ref.OracleRawGnn(g, torch.from_numpy(w['bag_words'] + 1), torch.from_numpy(w['bag_offsets']), int(w['counts'][3]), 16, 'gcn', 2, 1)
    m.pair_graph = pair_graph('small', 'uqi')
    m.load_reference_state({k[len('model.sd.'):]: z[k] for k in z.files if k.startswith('model.sd.')})
    u, q, i = (torch.from_numpy(z[f'model.{k}']) for k in 'uqi')
    scores = m(u, q, i)
    loss = torch.nn.BCEWithLogitsLoss()(scores, torch.from_numpy(z['model.flags']))
    loss.backward()
    assert rel_err(scores.detach().numpy(), z['model.scores']) <= RTOL and abs(loss.item() - float(z['model.loss'])) <= 1e-6
    for key, gr in m.reference_grads().items():
        assert rel_err(gr.numpy(), z[f'model.grad.{key}']) <= 5e-6, key
