"""CPU tests of the host side: file parsing, graph layout (native ihg_build_csr), samplers, metrics, schedule,
CLI - against the reference-generated fixtures.  No GPU compute is called anywhere in this file."""
import json
import os
import random

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from ihgnn_amd import synth
from ihgnn_amd.Dataset import GraphDataset, TestSearchLogDataLoader
from ihgnn_amd.Helpers.ArgsParser import parse_args
from ihgnn_amd.Helpers.Graph import PpsHyperGraph
from ihgnn_amd.Helpers.Metrics import Metrics, MetricsCollection
from ihgnn_amd.Helpers.ProcessController import ProcessController
from ihgnn_amd.Helpers.SearchLog import PosInteraction, SearchLog
from ihgnn_amd.layout import Csr, IncidenceLayout

CPU = torch.device('cpu')
F1 = os.path.join(GOLDEN, 'f1_data')


@pytest.fixture(scope='module')
def f1():
    z = np.load(os.path.join(GOLDEN, 'f1_graph.npz'))
    ds = GraphDataset(os.path.join(F1, 'graph_info.txt'), os.path.join(F1, 'queries_multihot.txt'),
                      os.path.join(F1, 'train_data.csv'), PpsHyperGraph, 10, 0, CPU)
    return z, ds


def test_dataset_parses_reference_files(f1):
    z, ds = f1
    assert [ds.user_count, ds.query_count, ds.item_count, ds.vocab_size, ds.node_count] == z['counts'].tolist()
    assert ds.query_start_index_in_graph == 5 and ds.item_start_index_in_graph == 9
    np.testing.assert_array_equal(ds.queries_for_embeddingbag.numpy(), z['bag_input'])
    np.testing.assert_array_equal(ds.queries_offset_for_embeddingbag.numpy(), z['bag_offsets'])
    np.testing.assert_array_equal(np.array([p.uqif() for p in ds.pos_interactions]), z['pos_uqif'])
    np.testing.assert_array_equal(np.array(ds.neg_interactions), z['neg_uqi'])
    np.testing.assert_array_equal(ds.users_onehot.numpy(), np.arange(1, 6))
    assert len(ds) == len(z['pos_uqif'])


def test_hypergraph_matches_reference(f1):
    z, ds = f1
    g = ds.hypergraph
    assert g is ds.graph and g.EdgeCount == int(z['EdgeCount'])
    np.testing.assert_array_equal(g.I3.numpy(), z['I3'])
    np.testing.assert_array_equal(g.Adjacency.indices().numpy(), z['coo_indices'])
    np.testing.assert_array_equal(g.Adjacency.values().numpy(), z['coo_values'])
    np.testing.assert_array_equal(g.VertexDegrees.numpy(), z['VertexDegrees'])
    np.testing.assert_array_equal(g.EdgeDegrees.numpy(), z['EdgeDegrees'])
    lay = g.layout
    # kernel-side scale factors: bit-identical to the reference's pow() where a node has edges, 0 where it has none
    deg = torch.from_numpy(z['VertexDegrees'][:, 0])
    alive = deg > 0.5
    assert torch.equal(lay.inv_deg[alive], deg.pow(-1)[alive]) and torch.equal(lay.inv_sqrt_deg[alive], deg.pow(-0.5)[alive])
    assert (lay.inv_deg[~alive] == 0).all() and (~alive).sum() == 2


def test_hypergraph_with_multiplicities_presents_the_reference_tensors(monkeypatch):
    """Fixture F1 holds a duplicated (user, query, item) triple (duplicates are distinct hyperedges).  With the layout collapsing it into one weighted row
    (``IHG_EDGE_MULTIPLICITY=1``) the graph object still presents the REFERENCE's tensors - ``EdgeCount``, ``I3``, ``Adjacency``, ``VertexDegrees``, ``EdgeDegrees`` in file
    order, one hyperedge per interaction - and the kernel-side scale factors are the same bits."""
    from ihgnn_amd import layout as layout_mod
    monkeypatch.setattr(layout_mod, 'EDGE_MULTIPLICITY', '1')
    z = np.load(os.path.join(GOLDEN, 'f1_graph.npz'))
    ds = GraphDataset(os.path.join(F1, 'graph_info.txt'), os.path.join(F1, 'queries_multihot.txt'), os.path.join(F1, 'train_data.csv'), PpsHyperGraph, 10, 0, CPU)
    g, lay = ds.hypergraph, ds.hypergraph.layout
    assert lay.edge_weight is not None and lay.edge_count == len(np.unique(z['I3'], axis=0)) < int(z['EdgeCount']) and float(lay.edge_weight.sum()) == int(z['EdgeCount']) and float(lay.edge_weight.max()) == 2.0
    assert g.EdgeCount == int(z['EdgeCount'])
    np.testing.assert_array_equal(g.I3.numpy(), z['I3'])
    np.testing.assert_array_equal(g.Adjacency.indices().numpy(), z['coo_indices'])
    np.testing.assert_array_equal(g.Adjacency.values().numpy(), z['coo_values'])
    np.testing.assert_array_equal(g.VertexDegrees.numpy(), z['VertexDegrees'])
    np.testing.assert_array_equal(g.EdgeDegrees.numpy(), z['EdgeDegrees'])
    deg = torch.from_numpy(z['VertexDegrees'][:, 0])
    alive = deg > 0.5
    assert torch.equal(lay.inv_deg[alive], deg.pow(-1)[alive]) and torch.equal(lay.inv_sqrt_deg[alive], deg.pow(-0.5)[alive]) and (lay.inv_deg[~alive] == 0).all()
    np.testing.assert_array_equal(lay.i3_host[lay.file_to_edge].astype(np.int64), z['I3'])


def test_csr_is_node_major_sorted_and_complete():
    w = synth.draw(50, 20, 70, 30, 2000, seed=3, distribution='powerlaw')
    lay = IncidenceLayout(w.triples, 50, 20, 70, CPU, heavy_threshold=64)
    assert (np.diff(lay.i3_host[:, 0]) >= 0).all() and np.array_equal(lay.i3_host[:, 0], w.triples[lay.edge_perm, 0])   # renumbered by user
    ptr, ids = lay.node_csr.ptr_host, lay.node_csr.ids_host
    i3 = lay.i3_host
    assert ptr[0] == 0 and ptr[-1] == 3 * 2000 and (np.diff(ptr) >= 0).all()
    for v in range(140):
        mine = ids[ptr[v]:ptr[v + 1]]
        assert (np.diff(mine) > 0).all()                        # ascending hyperedge ids, no duplicates
        assert all(v in i3[e] for e in mine)
    assert np.array_equal(np.bincount(i3.reshape(-1), minlength=140), np.diff(ptr))
    # split-row plan covers exactly the heavy rows, segment by segment
    csr = lay.node_csr
    assert csr.n_heavy == int((np.diff(ptr) > 64).sum()) > 0
    segptr = csr.heavy_segptr.numpy()
    for h, row in enumerate(csr.heavy_rows.numpy()):
        b, e = csr.seg_begin.numpy()[segptr[h]:segptr[h + 1]], csr.seg_end.numpy()[segptr[h]:segptr[h + 1]]
        assert b[0] == ptr[row] and e[-1] == ptr[row + 1] and (b[1:] == e[:-1]).all() and ((e - b) <= 128).all()


def test_member_lists_cut_by_hyperedge_range_partition_the_full_lists():
    """``member_csr_chunks`` (the interactive backward in pieces): every chunk's lists are the full list's entries of that
    hyperedge range, rebased; chunk boundaries are whole tiles; nothing is lost or repeated."""
    w = synth.draw(50, 20, 70, 30, 2000, seed=4, distribution='powerlaw')
    lay = IncidenceLayout(w.triples, 50, 20, 70, CPU, heavy_threshold=64)
    full_ptr, full_ids = lay.member_csr.ptr_host, lay.member_csr.ids_host
    chunks = lay.member_csr_chunks(3)
    assert [c[0] for c in chunks] == [0, 768, 1536] and chunks[-1][1] == 2000 and all(c[0] % 128 == 0 for c in chunks)
    assert lay.member_csr_chunks(3) is chunks                                    # built once
    for v in range(140):
        mine = full_ids[full_ptr[v]:full_ptr[v + 1]]
        pieces = [csr.ids_host[csr.ptr_host[v]:csr.ptr_host[v + 1]].astype(np.int64) + 3 * e0 for e0, _, csr in chunks]
        assert np.array_equal(np.concatenate(pieces), mine)
        for (e0, e1, _), piece in zip(chunks, pieces):
            assert ((piece // 3 >= e0) & (piece // 3 < e1)).all()


def test_split_row_plan_covers_every_long_row_once_and_caps_its_segments(monkeypatch):
    """Rows longer than the threshold are cut into segments of `heavy_chunk` ids - but never more than HEAVY_MAX_SEGMENTS per row (the extreme
    rows of a power-law graph get longer segments instead): the segments of a row tile its id range exactly, in order."""
    from ihgnn_amd import layout
    rng = np.random.default_rng(0)
    lens = np.concatenate([rng.integers(0, 40, 200), [257, 1000, 130_000, 5]])
    ptr = np.zeros(len(lens) + 1, np.int64)
    np.cumsum(lens, out=ptr[1:])
    ids = rng.integers(0, 1000, int(ptr[-1])).astype(np.int32)
    monkeypatch.setattr(layout, 'HEAVY_MAX_SEGMENTS', 64)
    csr = layout.Csr(ptr.astype(np.int32), ids, CPU, heavy_threshold=256, heavy_chunk=128)
    heavy = csr.heavy_rows.numpy()
    assert list(heavy) == [200, 201, 202]
    segptr, begin, end = csr.heavy_segptr.numpy(), csr.seg_begin.numpy(), csr.seg_end.numpy()
    for k, row in enumerate(heavy):
        b, e = begin[segptr[k]:segptr[k + 1]], end[segptr[k]:segptr[k + 1]]
        assert b[0] == ptr[row] and e[-1] == ptr[row + 1] and (b[1:] == e[:-1]).all() and (e > b).all()
        assert len(b) <= 64
    assert segptr[1] - segptr[0] == 3 and segptr[2] - segptr[1] == 8            # 257 and 1,000 ids in pieces of 128
    assert segptr[3] - segptr[2] == 64 and (end - begin).max() == -(-130_000 // 64)  # the long row: 64 longer pieces


def test_build_csr_rejects_bad_ids():
    from ihgnn_amd._lib import IhgnnHipError
    with pytest.raises(IhgnnHipError, match='out of range'):
        IncidenceLayout(np.array([[0, 0, 9]]), 2, 2, 3, CPU)


def test_transpose_csr():
    rng = np.random.default_rng(0)
    lens = rng.integers(0, 6, 40)
    ptr = np.zeros(41, np.int32); ptr[1:] = np.cumsum(lens)
    ids = rng.integers(0, 17, ptr[-1]).astype(np.int32)
    t = Csr(ptr, ids, CPU).transpose(17)
    dense = np.zeros((40, 17), np.int64)
    for r in range(40):
        for c in ids[ptr[r]:ptr[r + 1]]:
            dense[r, c] += 1
    for c in range(17):
        rows = t.ids_host[t.ptr_host[c]:t.ptr_host[c + 1]]
        assert (np.diff(rows) >= 0).all()
        assert np.array_equal(np.bincount(rows, minlength=40), dense[:, c])


def test_eval_loader_matches_reference(f1):
    z, ds = f1
    for key, fn in (('valid', 'valid_data.csv'), ('test', 'test_data.csv')):
        loader = TestSearchLogDataLoader(os.path.join(F1, fn), ds, CPU)
        assert np.array_equal(np.array([(l[0], l[1]) for l in loader.logs]), z[f'{key}_uq'])
        assert sum([l[2] for l in loader.logs], []) == z[f'{key}_items_flat'].tolist()
        for (users, queries, items, flags, all1), (u, q) in zip(loader, z[f'{key}_uq']):
            assert users.shape == (ds.item_count,) and (users == u).all() and (queries == q).all()
            assert flags is None and all1 is True


def test_getitem_and_collate_follow_reference_contract(f1):
    _, ds = f1
    ds10 = GraphDataset.from_arrays(5, 4, 60, 7, ds.bag_words_host - 1, ds.bag_offsets_host, ds.pos_triples, device=CPU)
    random.seed(5)
    (u, q, i, flag), negs = ds10[2]
    random.seed(5)
    assert negs == random.sample(range(60), 10) and len(set(negs)) == 10 and flag == 1
    assert (u, q, i) == tuple(ds.pos_triples[2])
    ds6 = GraphDataset.from_arrays(5, 4, 60, 7, ds.bag_words_host - 1, ds.bag_offsets_host, ds.pos_triples,
                                   random_negative_sample_size=3, device=CPU)
    batch = [ds6[k] for k in (0, 3, 5)]
    out = GraphDataset.collate_fn(batch)
    assert len(out) == 8 and all(t.dtype == torch.int64 for t in out)
    pu, pq, pi, pf, nu, nq, ni, nf = out
    assert pu.tolist() == [b[0][0] for b in batch] and pf.tolist() == [1, 1, 1]
    assert nu.tolist() == sum([[b[0][0]] * 3 for b in batch], []) and nq.tolist() == sum([[b[0][1]] * 3 for b in batch], [])
    assert ni.tolist() == sum([b[1] for b in batch], []) and nf.tolist() == [0] * 9


def test_logged_negative_sampling(f1):
    _, ds = f1
    ds2 = GraphDataset(os.path.join(F1, 'graph_info.txt'), os.path.join(F1, 'queries_multihot.txt'),
                       os.path.join(F1, 'train_data.csv'), PpsHyperGraph, 2, 1, CPU)
    assert ds2.neg_items_for_user_query_pair[(0, 0)] == [1] and ds2.neg_items_for_user_query_pair[(0, 1)] == [4]
    (_, _, _, _), negs = ds2[0]                # (0,0) has one logged negative -> it is taken, then 2 random
    assert negs[0] == 1 and len(negs) == 3
    (_, _, _, _), negs = ds2[1]                # (1,0) has none -> 3 random
    assert len(negs) == 3


def test_search_log_roundtrip_and_relevance_policy():
    row = '17,3,1400000000,42 7 42,1 1 1,1 2 3,2 0 1,1400000000 NA 1400000001'
    log = SearchLog.parse(row)
    assert (log.user, log.query, log.items, log.interactions) == (17, 3, [42, 7, 42], [2, 0, 1])
    assert log.tostr() == row
    assert log.get_interacted_items() == ([42], [1], True)
    assert log.get_interacted_items('max') == ([42], [2], False)
    pos = PosInteraction.from_search_log(log, treat_all_1=True)
    assert [p.uqif() for p in pos] == [(17, 3, 42, 1), (17, 3, 42, 1)]
    assert [p.uqif() for p in PosInteraction.from_search_log(log, False)] == [(17, 3, 42, 2), (17, 3, 42, 1)]
    with pytest.raises(ValueError):
        SearchLog.parse('1,2,3')


def test_metrics_known_answers():
    rec = json.load(open(os.path.join(GOLDEN, 'f4_metrics.json')))
    sc = rec['selfcheck']
    m = Metrics.calculate_on_all_items(torch.tensor(sc['scores']), sc['truth'], [1, 1, 2], True)
    assert m.to_string(no_title=True) == '1.0000 0.6653 0.5000'         # Helpers/Metrics.py __main__ self-check
    assert abs(Metrics._get_idcg_for_all1(3) - sc['idcg3']) < 1e-15 and abs(Metrics._get_idcg([2, 1, 1]) - sc['idcg_211']) < 1e-15
    for c in rec['random_cases']:
        m = Metrics.calculate_on_all_items(torch.tensor(c['scores']), c['truth'], None, True)
        assert (abs(m.HitRatio_at10 - c['hr']), abs(m.NDCG_at10 - c['ndcg']), abs(m.MAP_at10 - c['map'])) < (1e-12,) * 3
    c = rec['graded']
    m = Metrics.calculate_on_all_items(torch.tensor(c['scores']), c['truth'], c['flags'], False)
    assert (abs(m.HitRatio_at10 - c['hr']), abs(m.NDCG_at10 - c['ndcg']), abs(m.MAP_at10 - c['map'])) < (1e-12,) * 3
    m1 = Metrics.calculate_on_all_items(torch.tensor(sc['scores']), sc['truth'], None, True)
    col = MetricsCollection(True)
    col.add(10, m1, m1), col.add(20, m1.divide_and_get_new(0.5), m1.divide_and_get_new(0.5)), col.add(30, m1.divide_and_get_new(2), m1.divide_and_get_new(2))
    assert col.get_valid_best(key=lambda x: x.NDCG_at10)[0] == rec['best_valid_epoch'] == 20
    with pytest.raises(ValueError):
        col.add(40, m1)


@pytest.mark.parametrize('key,args', [('schedule_20_5_7_2', (20, 5, 7, 2)), ('schedule_12_1_3_3_store', (12, 1, 3, 3, 12, 1000000))])
def test_process_controller_schedule(key, args):
    rec = json.load(open(os.path.join(GOLDEN, 'f4_metrics.json')))
    pc = ProcessController(*args)
    got = [dict(epoch=e, test=pc.ShouldTest(), store=pc.ShouldStore()) for e in pc]
    assert got == rec[key] and len(pc) == args[0]


def test_cli_flags_and_aliases():
    a = parse_args(['--ds', 'Amazon/X/', '--gnn', 'hgcn', '--gnns', '3', '--fo', '2', '--emb', '64', '-c', '-m', '--ec', '7',
                    '--est', '2', '--etf', '1', '-d', '1', '--cp', 'latest'])
    assert (a.dataset, a.gnn, a.gnns, a.feature_order, a.embedding_size) == ('Amazon/X/', 'hgcn', 3, 2, 64)
    assert a.storecheckpoint and a.storemetrics and (a.epoch_count, a.epoch_start_test, a.epoch_test_frequency) == (7, 2, 1)
    assert a.device == '1' and a.checkpoint == 'latest'
    d = parse_args([])
    assert (d.epoch_count, d.gnns, d.feature_order, d.embedding_size, d.completeness, d.checkpoint) == (0, 0, 0, 0, 'uqi', '')
    from ihgnn_amd.Models import parse_gnn_layer, IHGNNLayer, HGCNLayer
    assert parse_gnn_layer['IHGNN'] is parse_gnn_layer['ihgnn'] is parse_gnn_layer['IHGNNLayer'] is IHGNNLayer
    assert parse_gnn_layer['hgcn'] is HGCNLayer and parse_gnn_layer[''] is None


def test_synthetic_workload_is_deterministic_and_file_roundtrips(tmp_path):
    a, b = synth.draw_config('C1'), synth.draw_config('C1')
    assert np.array_equal(a.triples, b.triples) and a.edge_count == 20000 and a.node_count == 2500
    w = synth.draw(12, 6, 15, 9, 80, seed=1, eval_logs=5)
    paths = synth.write_files(w, str(tmp_path))
    ds = GraphDataset(paths['fn_graph_info'], paths['fn_queries_multihot'], paths['fn_train_data'], PpsHyperGraph, 10, 0, CPU)
    assert np.array_equal(ds.pos_triples, w.triples) and np.array_equal(ds.bag_words_host - 1, w.bag_words)
    mem = GraphDataset.from_arrays(12, 6, 15, 9, w.bag_words, w.bag_offsets, w.triples, device=CPU)
    assert np.array_equal(mem.hypergraph.layout.i3_host, ds.hypergraph.layout.i3_host)
    p = synth.draw(1000, 100, 1000, 50, 20000, seed=2, distribution='powerlaw', exponent=1.2)
    deg = np.bincount(p.triples[:, 0], minlength=1000)
    assert deg.max() > 20 * np.median(deg[deg > 0])              # genuinely skewed


@pytest.mark.parametrize('tag,mode', [('tiny_uqi', 'uqi'), ('tiny_qi', 'qi'), ('small_uqi', 'uqi'), ('small_ui', 'ui')])
def test_pairwise_graph_matches_reference(f1, tag, mode):
    """Pps2DGraph via the native ihg_build_pair_csr: coalesced adjacency (duplicates summed) and degrees (fixture F7)."""
    from ihgnn_amd.Helpers.Graph import Pps2DGraph
    z = np.load(os.path.join(GOLDEN, 'f7_gcn.npz'))
    if tag.startswith('tiny'):
        triples, (U, Q, I) = f1[1].pos_triples, (5, 4, 6)
    else:
        w = np.load(os.path.join(GOLDEN, 'f2_small_workload.npz'))
        triples, (U, Q, I) = w['triples'], (40, 20, 50)
    g = Pps2DGraph.from_triples(triples, U + Q + I, U, Q, False, CPU, completeness=mode)
    np.testing.assert_array_equal(g.Adjacency.indices().numpy(), z[f'{tag}.adj_indices'])
    np.testing.assert_array_equal(g.Adjacency.values().numpy(), z[f'{tag}.adj_values'])
    np.testing.assert_array_equal(g.VertexDegrees.numpy(), z[f'{tag}.degrees'])
    looped = Pps2DGraph.from_triples(triples, U + Q + I, U, Q, True, CPU, completeness=mode)      # Graph.py:27-29
    assert torch.equal(looped.VertexDegrees[:, 0], torch.where(g.VertexDegrees[:, 0] < 0.5, torch.ones(U + Q + I), g.VertexDegrees[:, 0] + 1))
    assert looped.layout.csr.nnz == g.layout.csr.nnz + U + Q + I


def test_native_search_log_parser_matches_python_parser(tmp_path):
    """f4: ihg_parse_search_logs == SearchLog.parse + positive/negative split, on the hand-made fixture and a random file."""
    from ihgnn_amd.Dataset import parse_search_logs
    from ihgnn_amd.Helpers.SearchLogCollection import SearchLogCollection
    from ihgnn_amd._lib import IhgnnHipError
    rng = np.random.default_rng(3)
    fn = tmp_path / 'logs.csv'
    with open(fn, 'w') as f:
        f.write(synth.CSV_HEADER + '\n')
        for r in range(500):
            n = int(rng.integers(1, 7))
            items, flags = rng.integers(0, 1000, n), rng.integers(0, 3, n)
            f.write(f'{rng.integers(0, 50)},{rng.integers(0, 20)},1400000000,' + ' '.join(map(str, items)) + ',' + ' '.join(['1'] * n) + ',' +
                    ' '.join(map(str, range(n))) + ',' + ' '.join(map(str, flags)) + ',' + ' '.join(['NA'] * n) + '\n')
        f.write('\n')                                   # trailing blank line is ignored
    for path in (str(fn), os.path.join(F1, 'train_data.csv')):
        n_logs, pos, neg = parse_search_logs(path)
        logs = SearchLogCollection.read(path)
        want_pos = [(l.user, l.query, i) for l in logs for i, fl in zip(l.items, l.interactions) if fl > 0]
        want_neg = [(l.user, l.query, i) for l in logs for i, fl in zip(l.items, l.interactions) if fl <= 0]
        assert n_logs == len(logs) and pos.tolist() == [list(t) for t in want_pos] and neg.tolist() == [list(t) for t in want_neg]
    bad = tmp_path / 'bad.csv'
    bad.write_text(synth.CSV_HEADER + '\n1,2,3,4 5,1 1,1 2,1 0\n')          # 7 columns
    with pytest.raises(IhgnnHipError, match='line 2 has 7 columns'):
        parse_search_logs(str(bad))
    bad.write_text(synth.CSV_HEADER + '\n1,2,3,4 x,1 1,1 2,1 0,NA NA\n')
    with pytest.raises(IhgnnHipError, match='bad item'):
        parse_search_logs(str(bad))
    with pytest.raises(IhgnnHipError, match='cannot open'):
        parse_search_logs(str(tmp_path / 'missing.csv'))


def test_native_readers_and_parser_match_the_reference_outputs():
    """f4: the three native readers against what the REFERENCE's own loaders produced from the same files (fixtures F1 and F9:
    `pos_uqif`, `neg_uqi`, EmbeddingBag input / offsets, counts), not against this repo's Python parser."""
    from ihgnn_amd.Dataset import parse_search_logs, read_graph_info, read_query_bags
    for data, fixture in ((F1, 'f1_graph.npz'), (os.path.join(GOLDEN, 'f9_data'), 'f9_log_hypergraph.npz')):
        z = np.load(os.path.join(GOLDEN, fixture))
        n_logs, pos, neg, rows = parse_search_logs(os.path.join(data, 'train_data.csv'), with_rows=True)
        np.testing.assert_array_equal(pos, z['pos_uqif'][:, :3])
        np.testing.assert_array_equal(neg, z['neg_uqi'])
        assert rows.shape[0] == pos.shape[0] and (np.diff(rows) >= 0).all() and rows.max() < n_logs
        assert read_graph_info(os.path.join(data, 'graph_info.txt')) == [int(x) for x in z['counts'][:4]]
        words, offsets = read_query_bags(os.path.join(data, 'queries_multihot.txt'))
        np.testing.assert_array_equal(words + 1, z['bag_input'])           # Dataset.py:168: ids shifted past the padding row
        np.testing.assert_array_equal(offsets, z['bag_offsets'])


def test_query_bag_reader_edge_cases(tmp_path):
    from ihgnn_amd.Dataset import read_graph_info, read_query_bags
    from ihgnn_amd._lib import IhgnnHipError
    fn = tmp_path / 'q.txt'
    fn.write_text('3 1 2\n\n7\n')                                       # a query without words in the middle, newline at the end
    words, offsets = read_query_bags(str(fn))
    assert words.tolist() == [3, 1, 2, 7] and offsets.tolist() == [0, 3, 3]
    fn.write_text('3 1 2\n5')                                             # no trailing newline
    words, offsets = read_query_bags(str(fn))
    assert words.tolist() == [3, 1, 2, 5] and offsets.tolist() == [0, 3]
    fn.write_text('3 x\n')
    with pytest.raises(IhgnnHipError, match='bad word id'):
        read_query_bags(str(fn))
    info = tmp_path / 'g.txt'
    info.write_text('5 4 6\n')
    with pytest.raises(IhgnnHipError, match='users queries items vocabulary'):
        read_graph_info(str(info))


def test_log_hypergraph_matches_reference():
    """f4: the native per-search-log hypergraph builder against the reference's PpsLogHyperGraph tensors (fixture F9), bit for bit:
    coalesced incidence (one entry of value 2 for the repeated item), vertex degrees, edge degrees = arity."""
    from ihgnn_amd.Dataset import GraphDataset
    from ihgnn_amd.Helpers.Graph import PpsLogHyperGraph
    z = np.load(os.path.join(GOLDEN, 'f9_log_hypergraph.npz'))
    d = os.path.join(GOLDEN, 'f9_data')
    ds = GraphDataset(os.path.join(d, 'graph_info.txt'), os.path.join(d, 'queries_multihot.txt'), os.path.join(d, 'train_data.csv'),
                      PpsLogHyperGraph, 10, 0, CPU)
    g = ds.graph
    assert isinstance(g, PpsLogHyperGraph) and g.EdgeCount == int(z['EdgeCount'])
    np.testing.assert_array_equal(g.Adjacency.indices().numpy(), z['adj_indices'])
    np.testing.assert_array_equal(g.Adjacency.values().numpy(), z['adj_values'])
    np.testing.assert_array_equal(g.VertexDegrees.numpy(), z['VertexDegrees'])
    np.testing.assert_array_equal(g.EdgeDegrees.numpy(), z['EdgeDegrees'])
    lay = g.layout
    assert lay.edge_values is not None and float(lay.edge_values_host.max()) == 2.0
    # the two orientations describe the same matrix
    e_rows = np.repeat(np.arange(lay.edge_count), np.diff(lay.edge_csr.ptr_host))
    a = sorted(zip(lay.edge_csr.ids_host.tolist(), e_rows.tolist(), lay.edge_values_host.tolist()))
    n_rows = np.repeat(np.arange(lay.node_count), np.diff(lay.node_csr.ptr_host))
    b = sorted(zip(n_rows.tolist(), lay.node_csr.ids_host.tolist(), lay.node_values_host.tolist()))
    assert a == b
    # the entry point the reference's dataset calls
    g2 = PpsLogHyperGraph.from_search_logs(ds.search_logs, ds.node_count, ds.user_count, ds.query_count, CPU)
    np.testing.assert_array_equal(g2.Adjacency.indices().numpy(), z['adj_indices'])


def test_model_refuses_a_feature_width_the_evaluation_kernel_cannot_score_at_construction():
    """d (L + 1) beyond the scoring kernel's LDS pair block is refused when the model is BUILT - not at the first evaluation, an epoch of training later."""
    import torch
    from ihgnn_amd.Models import HemPredictionLayer, IHGNNLayer, RawGnn
    with pytest.raises(NotImplementedError, match='1280 exceeds 1264'):
        RawGnn(torch.device('cpu'), None, 256, IHGNNLayer, 4, 3, False, HemPredictionLayer, 0.5)


def test_readme_lists_every_switch_the_code_reads():
    """Every IHG_* / IHGNN_* environment variable the package or the library reads is a row of README.md's switch table (a switch whose A/B is settled leaves
    the code and the table together)."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    readme = open(os.path.join(root, 'README.md')).read()
    read = set()
    for path in glob.glob(os.path.join(root, 'ihgnn_amd', '**', '*.py'), recursive=True) + glob.glob(os.path.join(root, 'ihgnn_amd', 'csrc', '*.h*')):
        text = open(path).read()
        read.update(re.findall(r"environ(?:\.get)?\(\s*'(IHG[A-Z_]*_[A-Z0-9_]+)'", text))
        read.update(re.findall(r'getenv\(\s*"(IHG[A-Z_]*_[A-Z0-9_]+)"', text))
    assert 'IHG_INTERACT_ARITH' in read and 'IHG_NODE_TABLES' in read, read
    missing = sorted(name for name in read if name not in readme)
    assert not missing, f'not in README.md\'s switch table: {missing}'



def test_merged_two_hop_list_is_the_pairwise_graph_with_multiplicities():
    """``ihg_merge_id_lists`` on the two-hop list: per node the distinct other members of its hyperedges, ascending, with exact multiplicities - against numpy, and
    against the pairwise graph the reference builds for its GCN baseline (``Pps2DGraph``, completeness uqi: the same co-occurrence counts, ``Helpers/Graph.py:40-79``)."""
    import ctypes
    from ihgnn_amd import _lib
    from ihgnn_amd.layout import _as_ptr
    w = synth.draw(300, 12, 200, 10, 9000, seed=3, distribution='powerlaw', exponent=1.2)
    triples = np.concatenate([w.triples, w.triples[:40]])               # duplicate hyperedges stay distinct: their members count twice
    lay = IncidenceLayout(triples, 300, 12, 200, CPU)
    csr, weights, dup = lay.two_hop_merged()
    src = lay.hop2_csr
    assert csr.n_rows == src.n_rows and 0.05 < dup < 0.9 and abs((1 - dup) * src.nnz - csr.nnz) < 1
    sp, si = src.ptr_host.astype(np.int64), src.ids_host
    mp, mi, mw = csr.ptr_host.astype(np.int64), csr.ids_host, weights.numpy()
    for v in np.random.default_rng(0).integers(0, src.n_rows, 200).tolist() + [int(np.argmax(np.diff(sp)))]:
        ids, counts = np.unique(si[sp[v]:sp[v + 1]], return_counts=True)
        np.testing.assert_array_equal(mi[mp[v]:mp[v + 1]], ids)
        np.testing.assert_array_equal(mw[mp[v]:mp[v + 1]], counts.astype(np.float32))
    assert float(mw.sum()) == src.nnz
    # the pairwise graph of the same interactions (no self loops): identical rows, columns and values
    n = lay.node_count
    rowptr, cols, vals, degree = np.empty(n + 1, np.int32), np.empty(6 * len(triples), np.int32), np.empty(6 * len(triples), np.float32), np.empty(n, np.float32)
    nnz = ctypes.c_int64(0)
    _lib.check(_lib.load().ihg_build_pair_csr(_as_ptr(np.ascontiguousarray(triples, np.int64), ctypes.c_int64), len(triples), 300, 12, 200, 0, 0, _as_ptr(rowptr, ctypes.c_int32),
                                              _as_ptr(cols, ctypes.c_int32), _as_ptr(vals, ctypes.c_float), _as_ptr(degree, ctypes.c_float), 6 * len(triples), ctypes.byref(nnz)), 'pair')
    assert nnz.value == csr.nnz
    np.testing.assert_array_equal(rowptr, csr.ptr_host)
    np.testing.assert_array_equal(cols[:nnz.value], mi)
    np.testing.assert_array_equal(vals[:nnz.value], mw)
    # an empty graph and a graph of one hyperedge
    one = IncidenceLayout(np.array([[0, 0, 0]]), 2, 1, 1, CPU).two_hop_merged()
    assert one[0].nnz == 6 and one[2] == 0.0 and bool((one[1] == 1).all())


def test_unique_triples_and_weighted_layout():
    """``ihg_unique_triples`` + ``IncidenceLayout(edge_multiplicity='1')``: identical (user, query, item) triples kept once, ascending by (user, query, item), with their
    number of occurrences - against numpy; and every quantity the layout derives counts the copies exactly as the one-row-per-interaction layout does (the reference
    makes a hyperedge per interaction, duplicates included, ``Helpers/Graph.py:107-118``): degrees and scale factors bit for bit, the merged two-hop list entry for
    entry, the pair weights, the file -> row map."""
    from ihgnn_amd import layout as layout_mod
    w = synth.draw(300, 12, 200, 10, 9000, seed=3, distribution='powerlaw', exponent=1.2)
    triples = np.concatenate([w.triples, w.triples[:700], w.triples[:90]])       # on top of the draw's own repeats
    uniq, counts, where = layout_mod.unique_triples(triples, 300, 12, 200)
    want, want_inverse, want_counts = np.unique(triples, axis=0, return_inverse=True, return_counts=True)
    np.testing.assert_array_equal(uniq, want)
    np.testing.assert_array_equal(counts, want_counts.astype(np.float32))
    np.testing.assert_array_equal(where, want_inverse.reshape(-1))
    assert layout_mod.unique_triples(triples, 300, 12, 200, count_only=True) == len(want) < len(triples)
    with pytest.raises(Exception, match='out of range'):
        layout_mod.unique_triples(np.array([[0, 12, 0]]), 300, 12, 200)
    plain = IncidenceLayout(triples, 300, 12, 200, CPU, edge_multiplicity='0')
    lay = IncidenceLayout(triples, 300, 12, 200, CPU, edge_multiplicity='1')
    assert plain.edge_weight is None and plain.edge_count == plain.hyperedge_count == len(triples)
    assert lay.edge_count == len(want) and lay.hyperedge_count == len(triples) and abs(lay.duplicate_share - (1 - len(want) / len(triples))) < 1e-12
    np.testing.assert_array_equal(lay.edge_weight.numpy(), want_counts.astype(np.float32))
    np.testing.assert_array_equal(lay.file_to_edge, want_inverse.reshape(-1))
    np.testing.assert_array_equal(plain.file_to_edge[plain.edge_perm], np.arange(len(triples)))
    offs = np.array([0, 300, 312])
    np.testing.assert_array_equal(lay.i3_host, (want + offs).astype(np.int32))
    np.testing.assert_array_equal(lay.i3_host[lay.file_to_edge], (triples + offs).astype(np.int32))          # every interaction finds its row
    for name in ('degree', 'inv_deg', 'inv_sqrt_deg', 'self_weight'):
        assert torch.equal(getattr(lay, name), getattr(plain, name)), name
    # pair weights: one per incidence of a distinct hyperedge = that hyperedge's multiplicity
    np.testing.assert_array_equal(lay.pair_weight.numpy(), want_counts.astype(np.float32)[lay.node_csr.ids_host])
    assert lay.hop2_csr.nnz == 6 * len(want) and plain.pair_weight is None
    # the merged two-hop list (weights summed over the multiplicities) is the plain layout's merged list: H H^T - diag(deg) does not know how its entries were counted
    a, aw, a_dup = lay.two_hop_merged()
    b, bw, b_dup = plain.two_hop_merged()
    np.testing.assert_array_equal(a.ptr_host, b.ptr_host)
    np.testing.assert_array_equal(a.ids_host, b.ids_host)
    np.testing.assert_array_equal(aw.numpy(), bw.numpy())
    assert a_dup == b_dup == lay.two_hop_duplicate_share == plain.two_hop_duplicate_share
    assert lay.two_hop_merged_default and plain.two_hop_merged_default == (plain.two_hop_duplicate_share >= layout_mod.TWO_HOP_MERGED_MIN_SHARE)
    # auto: small graphs are left alone, large ones decide by the share of repeats
    assert IncidenceLayout(triples, 300, 12, 200, CPU).edge_weight is None
    big = np.concatenate([triples] * 8)
    auto = IncidenceLayout(big, 300, 12, 200, CPU)
    assert len(big) >= layout_mod.MULTIPLICITY_MIN_EDGES and auto.edge_weight is not None and auto.edge_count == len(want)
    np.testing.assert_array_equal(auto.edge_weight.numpy(), 8 * want_counts.astype(np.float32))
    sparse = synth.draw(30000, 3000, 30000, 10, 70000, seed=1)                   # uniform members: hardly a repeat
    assert IncidenceLayout(sparse.triples, 30000, 3000, 30000, CPU).edge_weight is None
    with pytest.raises(ValueError, match='edge_order'):
        IncidenceLayout(triples, 300, 12, 200, CPU, edge_order='file', edge_multiplicity='1')
    # the reference-shaped views keep presenting one hyperedge per interaction, in file order
    from ihgnn_amd.Helpers.Graph import PpsHyperGraph
    g = PpsHyperGraph()
    g.layout, g.EdgeCount = lay, lay.hyperedge_count
    assert tuple(g.I3.shape) == (len(triples), 3) and tuple(g.Adjacency.shape) == (512, len(triples)) and tuple(g.EdgeDegrees.shape) == (len(triples), 1)
    np.testing.assert_array_equal(g.I3.numpy(), triples + offs)


def test_split_row_plan_follows_the_lists_size():
    """Unset, the split-row plan is by the list's size (``layout.split_row_plan_for``): 256 / 128 below 2^20 entries - the graphs of tests and examples keep split rows on
    their path -, 512 / 512 from there on (every BASELINE config; ``profiles/r6/09_ab_split_row_plan.txt``); a caller's threshold or chunk always wins."""
    from ihgnn_amd import layout as layout_mod
    if layout_mod.HEAVY_THRESHOLD is None and layout_mod.HEAVY_CHUNK is None:
        assert layout_mod.split_row_plan_for(1000) == (256, 128) and layout_mod.split_row_plan_for((1 << 20) - 1) == (256, 128)
        assert layout_mod.split_row_plan_for(1 << 20) == (512, 512) and layout_mod.split_row_plan_for(300_000_000) == (512, 512)
        assert layout_mod.split_row_plan_for(10, 64) == (64, 128) and layout_mod.split_row_plan_for(1 << 22, 1024) == (1024, 512) and layout_mod.split_row_plan_for(1 << 22, 300, 100) == (300, 100)
    rng = np.random.default_rng(0)
    lens = np.concatenate([np.full(10, 700), rng.integers(0, 40, 5000)])
    ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    ids = rng.integers(0, 1000, int(ptr[-1])).astype(np.int32)
    small = layout_mod.Csr(ptr, ids, CPU)
    threshold, chunk = layout_mod.split_row_plan_for(int(ids.shape[0]))
    assert small.heavy_threshold == threshold and small.n_heavy == int((lens > threshold).sum()) and small.n_segments == int(sum(-(-l // chunk) for l in lens if l > threshold))
    derived = layout_mod.Csr(ptr, ids, CPU, small.heavy_threshold)          # (a list derived from another: its threshold, the chunk that goes with it)
    assert derived.n_segments == small.n_segments


def test_layout_without_the_isolated_nodes():
    """``IncidenceLayout(compact_nodes='1')``: the nodes that are in no hyperedge are left out of the layout's own numbering (their layer outputs are exactly zero, SURVEY
    App. B 2): the compact graph is the public one renamed through ``node_map`` / ``active_nodes``, every per-node array is the public one restricted to the active nodes,
    the reference-shaped views (``VertexDegrees`` with its 1e-8, ``I3``, ``Adjacency``) stay public.  auto: by the share of isolated nodes, large graphs only."""
    from ihgnn_amd import layout as layout_mod
    from ihgnn_amd.Helpers.Graph import PpsHyperGraph
    rng = np.random.default_rng(5)
    U, Q, I = 900, 40, 700
    live_u, live_i = rng.choice(U, 300, replace=False), rng.choice(I, 250, replace=False)
    triples = np.stack([rng.choice(live_u, 5000), rng.integers(0, Q, 5000), rng.choice(live_i, 5000)], 1)
    full = IncidenceLayout(triples, U, Q, I, CPU, compact_nodes='0', edge_multiplicity='0')
    lay = IncidenceLayout(triples, U, Q, I, CPU, compact_nodes='1', edge_multiplicity='0')
    assert not full.compact and full.node_map is None and lay.compact
    assert (lay.public_user_count, lay.public_query_count, lay.public_item_count, lay.public_node_count) == (U, Q, I, U + Q + I)
    alive = np.zeros(U + Q + I, bool)
    alive[np.unique(triples + np.array([0, U, U + Q]))] = True
    assert lay.node_count == int(alive.sum()) < U + Q + I and abs(lay.isolated_share - (1 - alive.mean())) < 1e-12
    assert lay.user_count == len(np.unique(triples[:, 0])) and lay.item_count == len(np.unique(triples[:, 2]))
    node_map, active = lay.node_map.numpy(), lay.active_nodes.numpy()
    np.testing.assert_array_equal(np.nonzero(alive)[0], active)
    np.testing.assert_array_equal(node_map[active], np.arange(lay.node_count))
    assert (node_map[~alive] == -1).all()
    # the same hypergraph, renamed: members, lists, degrees
    np.testing.assert_array_equal(active[lay.i3_host], full.i3_host)
    for name in ('degree', 'inv_deg', 'inv_sqrt_deg', 'self_weight'):
        assert torch.equal(getattr(lay, name), getattr(full, name)[lay.active_nodes]), name
    fp, cp = full.node_csr.ptr_host.astype(np.int64), lay.node_csr.ptr_host.astype(np.int64)
    np.testing.assert_array_equal(np.diff(cp), np.diff(fp)[active])
    np.testing.assert_array_equal(lay.node_csr.ids_host, full.node_csr.ids_host)
    np.testing.assert_array_equal(active[lay.hop2_csr.ids_host], full.hop2_csr.ids_host)
    assert torch.equal(lay.compact_rows(torch.tensor([int(active[3]), int(np.nonzero(~alive)[0][0])])), torch.tensor([3, -1]))
    assert torch.equal(lay.compact_rows(torch.tensor([int(np.nonzero(~alive)[0][0])]), isolated_to=0), torch.tensor([0]))
    # reference-shaped views stay public
    g = PpsHyperGraph()
    g.layout, g.EdgeCount = lay, lay.hyperedge_count
    want_deg = np.bincount((triples + np.array([0, U, U + Q])).reshape(-1), minlength=U + Q + I).astype(np.float32)
    want_deg[want_deg == 0] = 1e-8
    np.testing.assert_array_equal(g.VertexDegrees.numpy()[:, 0], want_deg)
    np.testing.assert_array_equal(g.I3.numpy(), triples + np.array([0, U, U + Q]))
    assert tuple(g.Adjacency.shape) == (U + Q + I, 5000)
    # with multiplicities on top
    both = IncidenceLayout(np.concatenate([triples, triples[:2000]]), U, Q, I, CPU, compact_nodes='1', edge_multiplicity='1')
    assert both.compact and both.edge_weight is not None and both.node_count == lay.node_count and both.hyperedge_count == 7000
    assert torch.equal(both.public_degree()[lay.active_nodes], both.degree) and float(both.degree.sum()) == 3 * 7000
    # auto: small graphs and graphs with few isolated nodes are left alone
    assert not IncidenceLayout(triples, U, Q, I, CPU).compact
    big = synth.draw(60000, 2000, 60000, 10, 40000, seed=2)                    # 122 k nodes, most of them never drawn
    assert IncidenceLayout(big.triples, 60000, 2000, 60000, CPU).compact
    dense = synth.draw(30000, 2000, 34000, 10, 600000, seed=2)
    auto = IncidenceLayout(dense.triples, 30000, 2000, 34000, CPU)
    assert not auto.compact and 0 <= auto.isolated_share < layout_mod.COMPACT_MIN_SHARE


# ---------------------------------------------------------------------------------------------
# the operand splits of the contraction kernels, emulated (tests/split_emulation.py; csrc/split_common.hpp)
# ---------------------------------------------------------------------------------------------
def test_two_fp16_split_worst_case_significand_and_bound():
    """Search of all 2^23 significands: the value whose products lose most under the two-fp16 round-to-nearest split with every omitted term of one sign is
    1 + 4093 * 2^-23 (hi = 1, lo = 4092 * 2^-23 by a tie to even, residual 2^-23): 2^-21 per product, one-sided.  The pattern the three-bf16 test used (low 16 bits
    set) has NO error at all under this split - which is why the GPU test takes its operands from this search now."""
    import split_emulation as se
    x, loss = se.worst_two_fp16_significand()
    assert x == np.float32(1.0 + 4093 * 2.0 ** -23)
    assert 0.99 * 2.0 ** -21 <= loss <= 2.0 ** -21
    old = np.array([(1.0 + (2.0 ** 16 - 1) * 2.0 ** -23) * 8192.0], np.float32)
    hi, lo = se.split_two_fp16(old)
    assert float(old[0]) - hi[0] - lo[0] == 0.0 and abs(lo[0] / hi[0]) < 2.0 ** -22          # exact, and lo*lo is 2^-45 of the product
    # a dot product of such values (all products positive): the loss adds up, it does not average out
    rng = np.random.default_rng(0)
    a = np.float32(x) * np.exp2(rng.integers(-2, 3, (4, 1024))).astype(np.float32)
    b = np.float32(x) * np.exp2(rng.integers(-2, 3, (3, 1024))).astype(np.float32)
    exact = a.astype(np.float64) @ b.astype(np.float64).T
    err = np.abs(se.dot_two_fp16(a, b) / exact - 1)
    assert 0.9 * 2.0 ** -21 <= err.min() and err.max() <= 3 * 2.0 ** -22                    # the bound split_common.hpp states
    assert np.abs(se.dot_three_bf16(a, b) / exact - 1).max() < 1e-8                          # harmless for the other scheme ...
    y, loss3 = se.worst_three_bf16_significand()
    assert y == np.float32(1.0 + (2.0 ** 16 - 1) * 2.0 ** -23) and 0.95 * 2.0 ** -21 <= loss3 <= 2.0 ** -21     # ... whose own worst case is the old pattern


def test_two_fp16_split_bound_on_random_and_wide_range_rows():
    """|error| <= 3 * 2^-22 * sum |a_k b_k| for rows whose entries lie within 2^-17 of the row's largest (lo is a normal fp16 there); on random data both schemes sit at
    ~ 2e-8.  Below 2^-17 of the row's largest entry lo is a subnormal: ABSOLUTE error 2^-25 scaled, i.e. 2^-38 of the row's largest entry per element - the scheme's
    bound is norm-wise there (|error| <= 2^-37 |a|_inf |b|_1 + ...), not component-wise; the one-huge-many-tiny case below states what that means."""
    import split_emulation as se
    rng = np.random.default_rng(1)
    for spread in (0, 8, 16):
        a = (rng.standard_normal((32, 512)) * np.exp2(rng.integers(-spread, 1, (32, 512)))).astype(np.float32)
        b = (rng.standard_normal((32, 512)) * np.exp2(rng.integers(-spread, 1, (32, 512)))).astype(np.float32)
        exact = a.astype(np.float64) @ b.astype(np.float64).T
        den = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64).T
        assert (np.abs(se.dot_two_fp16(a, b) - exact) / den).max() <= 3 * 2.0 ** -22, spread
        assert (np.abs(se.dot_three_bf16(a, b) - exact) / den).max() <= 2.0 ** -21, spread
    # one entry 2^24 above the rest, and the weight it meets is zero: the result is carried by entries whose lo term is subnormal
    a = rng.standard_normal((8, 256)).astype(np.float32)
    a[:, 0] = 2.0 ** 24
    b = rng.standard_normal((8, 256)).astype(np.float32)
    b[:, 0] = 0.0
    exact = a.astype(np.float64) @ b.astype(np.float64).T
    got = se.dot_two_fp16(a, b)
    norm_wise = np.abs(a).max(1)[:, None] * np.abs(b).sum(1)[None, :]
    assert (np.abs(got - exact) / norm_wise).max() <= 2.0 ** -36                             # tiny against |a|_inf |b|_1 ...
    component_wise = np.abs(got - exact) / (np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64).T)
    assert component_wise.max() > 1e-5                                                       # ... and NOT against sum |a_k b_k|: the stated limit of a per-row scale


def test_user_reduced_member_lists_in_chunks_are_cut_where_the_user_changes():
    """``IncidenceLayout.member_csr_qi_chunks``: the query / item member lists of the user-reduced backward cut by hyperedge range for buffers beyond
    ``ops.MEMBER_BUFFER_LIMIT_BYTES`` (config C5).  Every cut sits between two users (hyperedges are numbered by user), the chunks' lists are a partition of
    ``member_csr_qi``'s entries with ids rebased to the chunk, user rows are empty, and one user owning most hyperedges gives fewer, uneven chunks - never a cut
    inside its run."""
    w = synth.draw(300, 12, 200, 10, 9000, seed=5, distribution='powerlaw', exponent=1.2)
    lay = IncidenceLayout(w.triples, 300, 12, 200, CPU, edge_order='user')
    assert lay.user_sorted
    whole, _ = lay.member_csr_qi()
    wp, wi = whole.ptr_host.astype(np.int64), whole.ids_host.astype(np.int64)
    for n_chunks in (2, 3, 7):
        parts = lay.member_csr_qi_chunks(n_chunks)
        assert parts[0][0] == 0 and parts[-1][1] == lay.edge_count and 2 <= len(parts) <= n_chunks
        users = lay.i3_host[:, 0]
        gathered = [[] for _ in range(lay.node_count)]
        for (e0, e1, csr, rows), nxt in zip(parts, parts[1:] + [None]):
            assert e0 < e1 and (nxt is None or (nxt[0] == e1 and users[e1 - 1] != users[e1]))
            p, ids = csr.ptr_host.astype(np.int64), csr.ids_host.astype(np.int64)
            assert p[lay.user_count] == 0 and ids.size == 2 * (e1 - e0) and (ids.size == 0 or (ids.min() >= 0 and ids.max() < 2 * (e1 - e0)))
            np.testing.assert_array_equal(np.sort(rows.numpy()), np.arange(lay.user_count, lay.node_count))
            lens = np.diff(p)[rows.numpy()]
            assert (np.diff(lens) <= 0).all()                            # longest list first
            for v in range(lay.user_count, lay.node_count):
                gathered[v].append(ids[p[v]:p[v + 1]] + 2 * e0)
        for v in range(lay.user_count, lay.node_count):
            np.testing.assert_array_equal(np.sort(np.concatenate(gathered[v])), np.sort(wi[wp[v]:wp[v + 1]]))
    # one user in 90 % of the hyperedges: its run is never cut
    t = w.triples.copy()
    t[:8100, 0] = 7
    heavy = IncidenceLayout(t, 300, 12, 200, CPU, edge_order='user')
    parts = heavy.member_csr_qi_chunks(4)
    users = heavy.i3_host[:, 0]
    assert all(users[e1 - 1] != users[e1] for _, e1, _, _ in parts[:-1]) and sum(e1 - e0 for e0, e1, _, _ in parts) == heavy.edge_count
