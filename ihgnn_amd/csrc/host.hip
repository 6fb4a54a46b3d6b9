// host.cpp (compiled by hipcc with the rest) - host-side entry points: ABI version, error string, hypergraph / pairwise-graph
// layout builders, CSR transpose, search-log CSV ingestion.
#include "common.hpp"

#include <thread>

thread_local char ihg_error_buffer[512] = "";

namespace {

// Parses the space-separated integers of one CSV field into `out`; returns false on a malformed token.
bool parse_int_list(const char* p, const char* end, std::vector<int64_t>& out) {
    out.clear();
    while (p < end) {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
        if (p >= end) break;
        bool neg = false;
        if (*p == '-' || *p == '+') { neg = *p == '-'; ++p; }
        if (p >= end || *p < '0' || *p > '9') return false;
        int64_t v = 0;
        while (p < end && *p >= '0' && *p <= '9') v = v * 10 + (*p++ - '0');
        out.push_back(neg ? -v : v);
    }
    return true;
}

}  // namespace

extern "C" {

int32_t ihg_abi_version(void) { return 33; }

const char* ihg_last_error_string(void) { return ihg_error_buffer; }

int ihg_build_csr(const int64_t* triples, int64_t n_edges, int64_t n_users, int64_t n_queries, int64_t n_items,
                  int32_t* i3, int32_t* rowptr, int32_t* edge_ids, float* degree) {
    if (n_edges < 0 || n_users < 0 || n_queries < 0 || n_items < 0) return fail(IHG_ERR_INVALID, "ihg_build_csr: negative size");
    const int64_t n_nodes = n_users + n_queries + n_items;
    if (n_nodes >= INT32_MAX || n_edges * 3 >= INT32_MAX) return fail(IHG_ERR_INVALID, "ihg_build_csr: graph exceeds int32 indexing");
    if ((n_edges > 0 && (triples == nullptr || i3 == nullptr || edge_ids == nullptr)) || rowptr == nullptr || degree == nullptr)
        return fail(IHG_ERR_INVALID, "ihg_build_csr: null buffer");
    const int64_t offset[3] = {0, n_users, n_users + n_queries};
    const int64_t limit[3] = {n_users, n_queries, n_items};
    std::memset(rowptr, 0, sizeof(int32_t) * static_cast<size_t>(n_nodes + 1));
    for (int64_t e = 0; e < n_edges; ++e) {
        for (int m = 0; m < 3; ++m) {
            const int64_t local = triples[e * 3 + m];
            if (local < 0 || local >= limit[m])
                return fail(IHG_ERR_INVALID, "ihg_build_csr: hyperedge %lld member %d id %lld out of range [0,%lld)",
                            static_cast<long long>(e), m, static_cast<long long>(local), static_cast<long long>(limit[m]));
            const int32_t node = static_cast<int32_t>(local + offset[m]);
            i3[e * 3 + m] = node;
            ++rowptr[node + 1];
        }
    }
    for (int64_t v = 0; v < n_nodes; ++v) {
        const int32_t d = rowptr[v + 1];
        degree[v] = d == 0 ? 1e-8f : static_cast<float>(d);
        rowptr[v + 1] = rowptr[v] + d;
    }
    std::vector<int32_t> cursor(rowptr, rowptr + n_nodes);
    for (int64_t e = 0; e < n_edges; ++e)           // ascending e => ascending hyperedge ids inside every node
        for (int m = 0; m < 3; ++m) edge_ids[cursor[i3[e * 3 + m]]++] = static_cast<int32_t>(e);
    return IHG_OK;
}


int ihg_build_pair_csr(const int64_t* triples, int64_t n_edges, int64_t n_users, int64_t n_queries, int64_t n_items,
                       int32_t completeness, int32_t self_loops, int32_t* rowptr, int32_t* cols, float* vals, float* degree,
                       int64_t capacity, int64_t* nnz_out) {
    if (n_edges < 0 || n_users < 0 || n_queries < 0 || n_items < 0 || completeness < 0 || completeness > 3)
        return fail(IHG_ERR_INVALID, "ihg_build_pair_csr: bad argument");
    const int64_t n_nodes = n_users + n_queries + n_items;
    if (rowptr == nullptr || degree == nullptr || nnz_out == nullptr || (n_edges > 0 && (triples == nullptr || cols == nullptr || vals == nullptr)))
        return fail(IHG_ERR_INVALID, "ihg_build_pair_csr: null buffer");
    if (n_nodes >= INT32_MAX) return fail(IHG_ERR_INVALID, "ihg_build_pair_csr: graph exceeds int32 indexing");
    // member pairs joined by one interaction (Helpers/Graph.py:40-63): uqi = all three pairs, otherwise a single pair
    static const int kPairs[4][3][2] = {{{0, 1}, {1, 2}, {2, 0}}, {{0, 1}, {0, 1}, {0, 1}}, {{0, 2}, {0, 2}, {0, 2}}, {{1, 2}, {1, 2}, {1, 2}}};
    const int n_pairs = completeness == 0 ? 3 : 1;
    const int64_t offset[3] = {0, n_users, n_users + n_queries};
    const int64_t limit[3] = {n_users, n_queries, n_items};
    std::vector<uint64_t> keys;
    keys.reserve(static_cast<size_t>(n_edges) * n_pairs * 2 + (self_loops ? n_nodes : 0));
    std::vector<float> deg(static_cast<size_t>(n_nodes), self_loops ? 1.f : 0.f);
    if (self_loops)
        for (int64_t v = 0; v < n_nodes; ++v) keys.push_back((static_cast<uint64_t>(v) << 32) | static_cast<uint64_t>(v));
    for (int64_t e = 0; e < n_edges; ++e) {
        int64_t node[3];
        for (int m = 0; m < 3; ++m) {
            const int64_t local = triples[e * 3 + m];
            if (local < 0 || local >= limit[m]) return fail(IHG_ERR_INVALID, "ihg_build_pair_csr: interaction %lld member %d out of range", (long long)e, m);
            node[m] = local + offset[m];
        }
        for (int k = 0; k < n_pairs; ++k) {
            const uint64_t a = static_cast<uint64_t>(node[kPairs[completeness][k][0]]), b = static_cast<uint64_t>(node[kPairs[completeness][k][1]]);
            keys.push_back((a << 32) | b);
            keys.push_back((b << 32) | a);
            deg[a] += 1.f;                                   // Graph.py:48,54,60,66: +2 per member for uqi, +1 per pair member otherwise
            deg[b] += 1.f;
        }
    }
    std::sort(keys.begin(), keys.end());
    std::memset(rowptr, 0, sizeof(int32_t) * static_cast<size_t>(n_nodes + 1));
    int64_t nnz = 0;
    for (size_t k = 0; k < keys.size();) {
        size_t j = k;
        while (j < keys.size() && keys[j] == keys[k]) ++j;   // duplicates are summed by coalesce() (Graph.py:73-79)
        if (nnz >= capacity) return fail(IHG_ERR_WORKSPACE, "ihg_build_pair_csr: capacity %lld too small", (long long)capacity);
        cols[nnz] = static_cast<int32_t>(keys[k] & 0xffffffffu);
        vals[nnz] = static_cast<float>(j - k);
        ++rowptr[(keys[k] >> 32) + 1];
        ++nnz;
        k = j;
    }
    for (int64_t v = 0; v < n_nodes; ++v) {
        rowptr[v + 1] += rowptr[v];
        degree[v] = (!self_loops && deg[v] == 0.f) ? 1e-8f : deg[v];
    }
    *nnz_out = nnz;
    return IHG_OK;
}


// ------------------------------------------------------------------------------------------------
// HOST: search-log CSV ingestion.
// ------------------------------------------------------------------------------------------------

int ihg_parse_search_logs(const char* path, int64_t* n_logs, int64_t* n_pos, int64_t* n_neg, int64_t* pos, int64_t pos_capacity,
                          int64_t* neg, int64_t neg_capacity, int64_t* pos_log) {
    if (path == nullptr || n_logs == nullptr || n_pos == nullptr || n_neg == nullptr) return fail(IHG_ERR_INVALID, "ihg_parse_search_logs: null argument");
    FILE* f = std::fopen(path, "rb");
    if (f == nullptr) return fail(IHG_ERR_INVALID, "ihg_parse_search_logs: cannot open %s", path);
    std::fseek(f, 0, SEEK_END);
    const long size = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<char> text(static_cast<size_t>(size > 0 ? size : 0) + 1);
    const size_t got = size > 0 ? std::fread(text.data(), 1, static_cast<size_t>(size), f) : 0;
    std::fclose(f);
    text[got] = '\n';
    const char* p = text.data();
    const char* const end = p + got + 1;
    while (p < end && *p != '\n') ++p;                      // header line (SearchLogCollection.py:28)
    ++p;
    int64_t logs = 0, pcount = 0, ncount = 0, line_no = 1;
    std::vector<int64_t> items, flags, scalar;
    while (p < end) {
        const char* eol = p;
        while (eol < end && *eol != '\n') ++eol;
        ++line_no;
        const char* q = p;
        bool blank = true;
        for (const char* c = p; c < eol; ++c)
            if (*c != ' ' && *c != '\r' && *c != '\t') { blank = false; break; }
        if (!blank) {
            const char* field[9];
            int n_fields = 0;
            field[n_fields++] = q;
            for (const char* c = q; c < eol && n_fields < 9; ++c)
                if (*c == ',') field[n_fields++] = c + 1;
            int commas = 0;
            for (const char* c = q; c < eol; ++c) commas += *c == ',';
            if (commas != 7) return fail(IHG_ERR_INVALID, "ihg_parse_search_logs: %s line %lld has %d columns, expected 8", path, (long long)line_no, commas + 1);
            field[8] = eol + 1;
            auto fend = [&](int k) { return field[k + 1] - 1; };
            if (!parse_int_list(field[0], fend(0), scalar) || scalar.size() != 1) return fail(IHG_ERR_INVALID, "ihg_parse_search_logs: %s line %lld: bad user id", path, (long long)line_no);
            const int64_t user = scalar[0];
            if (!parse_int_list(field[1], fend(1), scalar) || scalar.size() != 1) return fail(IHG_ERR_INVALID, "ihg_parse_search_logs: %s line %lld: bad query id", path, (long long)line_no);
            const int64_t query = scalar[0];
            if (!parse_int_list(field[3], fend(3), items) || !parse_int_list(field[6], fend(6), flags))
                return fail(IHG_ERR_INVALID, "ihg_parse_search_logs: %s line %lld: bad item / interaction list", path, (long long)line_no);
            const size_t n = items.size() < flags.size() ? items.size() : flags.size();
            for (size_t k = 0; k < n; ++k) {
                if (flags[k] > 0) {
                    if (pos != nullptr) {
                        if (pcount >= pos_capacity) return fail(IHG_ERR_WORKSPACE, "ihg_parse_search_logs: positive buffer too small");
                        pos[pcount * 3] = user; pos[pcount * 3 + 1] = query; pos[pcount * 3 + 2] = items[k];
                        if (pos_log != nullptr) pos_log[pcount] = logs;
                    }
                    ++pcount;
                } else {
                    if (neg != nullptr) {
                        if (ncount >= neg_capacity) return fail(IHG_ERR_WORKSPACE, "ihg_parse_search_logs: negative buffer too small");
                        neg[ncount * 3] = user; neg[ncount * 3 + 1] = query; neg[ncount * 3 + 2] = items[k];
                    }
                    ++ncount;
                }
            }
            ++logs;
        }
        p = eol + 1;
    }
    *n_logs = logs;
    *n_pos = pcount;
    *n_neg = ncount;
    return IHG_OK;
}

// ------------------------------------------------------------------------------------------------
// HOST: the two small text files of a corpus.
// ------------------------------------------------------------------------------------------------
namespace {
bool slurp(const char* path, std::vector<char>& text) {
    FILE* f = std::fopen(path, "rb");
    if (f == nullptr) return false;
    std::fseek(f, 0, SEEK_END);
    const long size = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    text.resize(static_cast<size_t>(size > 0 ? size : 0) + 1);
    const size_t got = size > 0 ? std::fread(text.data(), 1, static_cast<size_t>(size), f) : 0;
    std::fclose(f);
    text[got] = '\n';
    text.resize(got + 1);
    return true;
}
}  // namespace

int ihg_read_graph_info(const char* path, int64_t* counts) {
    if (path == nullptr || counts == nullptr) return fail(IHG_ERR_INVALID, "ihg_read_graph_info: null argument");
    std::vector<char> text;
    if (!slurp(path, text)) return fail(IHG_ERR_INVALID, "ihg_read_graph_info: cannot open %s", path);
    const char* p = text.data();
    const char* eol = p;
    while (*eol != '\n') ++eol;
    std::vector<int64_t> v;
    if (!parse_int_list(p, eol, v) || v.size() != 4)
        return fail(IHG_ERR_INVALID, "ihg_read_graph_info: %s: expected \"users queries items vocabulary\" on the first line", path);
    for (int k = 0; k < 4; ++k) {
        if (v[k] < 0) return fail(IHG_ERR_INVALID, "ihg_read_graph_info: %s: negative count", path);
        counts[k] = v[k];
    }
    return IHG_OK;
}

int ihg_read_query_bags(const char* path, int64_t* n_queries, int64_t* n_words, int64_t* offsets, int64_t offsets_capacity, int64_t* words,
                        int64_t words_capacity) {
    if (path == nullptr || n_queries == nullptr || n_words == nullptr) return fail(IHG_ERR_INVALID, "ihg_read_query_bags: null argument");
    std::vector<char> text;
    if (!slurp(path, text)) return fail(IHG_ERR_INVALID, "ihg_read_query_bags: cannot open %s", path);
    const char* p = text.data();
    const char* const end = p + text.size();
    if (text.size() >= 2 && text[text.size() - 2] == '\n') {}         // file ends with a newline: the sentinel added by slurp is an empty tail
    int64_t q = 0, w = 0;
    std::vector<int64_t> line;
    while (p < end) {
        const char* eol = p;
        while (eol < end && *eol != '\n') ++eol;
        if (eol + 1 >= end && eol == p) break;                         // the empty tail behind the last newline is not a query
        if (!parse_int_list(p, eol, line)) return fail(IHG_ERR_INVALID, "ihg_read_query_bags: %s line %lld: bad word id", path, (long long)(q + 1));
        if (offsets != nullptr) {
            if (q >= offsets_capacity) return fail(IHG_ERR_WORKSPACE, "ihg_read_query_bags: offsets buffer too small");
            offsets[q] = w;
        }
        if (words != nullptr) {
            if (w + static_cast<int64_t>(line.size()) > words_capacity) return fail(IHG_ERR_WORKSPACE, "ihg_read_query_bags: words buffer too small");
            for (size_t k = 0; k < line.size(); ++k) words[w + static_cast<int64_t>(k)] = line[k];
        }
        w += static_cast<int64_t>(line.size());
        ++q;
        p = eol + 1;
    }
    *n_queries = q;
    *n_words = w;
    return IHG_OK;
}

// ------------------------------------------------------------------------------------------------
// HOST: per-search-log hypergraph (variable arity).
// ------------------------------------------------------------------------------------------------
int ihg_build_log_hypergraph(const int64_t* pos, const int64_t* pos_log, int64_t n_pos, int64_t n_users, int64_t n_queries, int64_t n_items,
                             int32_t* edge_ptr, int32_t* edge_nodes, float* edge_vals, float* edge_degree, int32_t* node_ptr,
                             int32_t* node_edges, float* node_vals, float* node_degree, int64_t* n_edges_out, int64_t* nnz_out) {
    if (n_pos < 0 || n_users < 0 || n_queries < 0 || n_items < 0) return fail(IHG_ERR_INVALID, "ihg_build_log_hypergraph: negative size");
    const int64_t n_nodes = n_users + n_queries + n_items;
    if (n_nodes >= INT32_MAX || n_pos * 3 >= INT32_MAX) return fail(IHG_ERR_INVALID, "ihg_build_log_hypergraph: graph exceeds int32 indexing");
    if (edge_ptr == nullptr || node_ptr == nullptr || node_degree == nullptr || n_edges_out == nullptr || nnz_out == nullptr ||
        (n_pos > 0 && (pos == nullptr || pos_log == nullptr || edge_nodes == nullptr || edge_vals == nullptr || edge_degree == nullptr ||
                       node_edges == nullptr || node_vals == nullptr)))
        return fail(IHG_ERR_INVALID, "ihg_build_log_hypergraph: null buffer");
    int64_t n_edges = 0, nnz = 0;
    edge_ptr[0] = 0;
    std::vector<int32_t> members;
    std::memset(node_ptr, 0, sizeof(int32_t) * static_cast<size_t>(n_nodes + 1));
    for (int64_t k = 0; k < n_pos;) {
        int64_t j = k;
        while (j < n_pos && pos_log[j] == pos_log[k]) ++j;             // positives of one search log are consecutive
        const int64_t u = pos[k * 3], q = pos[k * 3 + 1];
        if (u < 0 || u >= n_users || q < 0 || q >= n_queries)
            return fail(IHG_ERR_INVALID, "ihg_build_log_hypergraph: log %lld: user / query id out of range", (long long)pos_log[k]);
        members.clear();
        for (int64_t t = k; t < j; ++t) {
            const int64_t item = pos[t * 3 + 2];
            if (pos[t * 3] != u || pos[t * 3 + 1] != q) return fail(IHG_ERR_INVALID, "ihg_build_log_hypergraph: log %lld changes its user / query", (long long)pos_log[k]);
            if (item < 0 || item >= n_items) return fail(IHG_ERR_INVALID, "ihg_build_log_hypergraph: log %lld: item id out of range", (long long)pos_log[k]);
            members.push_back(static_cast<int32_t>(item + n_users + n_queries));
        }
        std::sort(members.begin(), members.end());
        // edge = [u, q + U, items ...] (Graph.py:160-162); a repeated item is one entry of value 2 after coalesce() (Graph.py:178-184)
        edge_nodes[nnz] = static_cast<int32_t>(u); edge_vals[nnz++] = 1.f;
        edge_nodes[nnz] = static_cast<int32_t>(q + n_users); edge_vals[nnz++] = 1.f;
        for (size_t t = 0; t < members.size();) {
            size_t r = t;
            while (r < members.size() && members[r] == members[t]) ++r;
            edge_nodes[nnz] = members[t];
            edge_vals[nnz++] = static_cast<float>(r - t);
            t = r;
        }
        edge_degree[n_edges] = static_cast<float>(2 + members.size());     // len(nodes), repeats counted (Graph.py:167)
        for (int64_t t = edge_ptr[n_edges]; t < nnz; ++t) ++node_ptr[edge_nodes[t] + 1];
        edge_ptr[++n_edges] = static_cast<int32_t>(nnz);
        k = j;
    }
    for (int64_t v = 0; v < n_nodes; ++v) {
        const int32_t d = node_ptr[v + 1];                             // vertex_degrees[nodes] += 1: once per edge the node is in (Graph.py:166)
        node_degree[v] = d == 0 ? 1e-8f : static_cast<float>(d);
        node_ptr[v + 1] = node_ptr[v] + d;
    }
    std::vector<int32_t> cursor(node_ptr, node_ptr + n_nodes);
    for (int64_t e = 0; e < n_edges; ++e)
        for (int32_t t = edge_ptr[e]; t < edge_ptr[e + 1]; ++t) {
            const int32_t slot = cursor[edge_nodes[t]]++;
            node_edges[slot] = static_cast<int32_t>(e);
            node_vals[slot] = edge_vals[t];
        }
    *n_edges_out = n_edges;
    *nnz_out = nnz;
    return IHG_OK;
}

int ihg_transpose_csr(const int32_t* ptr, const int32_t* ids, int64_t n_rows, int64_t n_cols, int32_t* t_ptr, int32_t* t_rows) {
    if (n_rows < 0 || n_cols < 0 || ptr == nullptr || t_ptr == nullptr) return fail(IHG_ERR_INVALID, "ihg_transpose_csr: bad argument");
    const int64_t nnz = ptr[n_rows];
    if (nnz > 0 && (ids == nullptr || t_rows == nullptr)) return fail(IHG_ERR_INVALID, "ihg_transpose_csr: null buffer");
    std::memset(t_ptr, 0, sizeof(int32_t) * static_cast<size_t>(n_cols + 1));
    for (int64_t k = 0; k < nnz; ++k) {
        if (ids[k] < 0 || ids[k] >= n_cols) return fail(IHG_ERR_INVALID, "ihg_transpose_csr: id %d out of range", ids[k]);
        ++t_ptr[ids[k] + 1];
    }
    for (int64_t c = 0; c < n_cols; ++c) t_ptr[c + 1] += t_ptr[c];
    std::vector<int32_t> cursor(t_ptr, t_ptr + n_cols);
    for (int64_t r = 0; r < n_rows; ++r)
        for (int32_t k = ptr[r]; k < ptr[r + 1]; ++k) t_rows[cursor[ids[k]]++] = static_cast<int32_t>(r);
    return IHG_OK;
}

int ihg_merge_id_lists(const int32_t* ptr, const int32_t* ids, const float* weights, int64_t n_rows, int32_t* out_ptr, int32_t* out_ids, float* out_counts,
                       int64_t* nnz_out) {
    if (n_rows < 0 || ptr == nullptr || out_ptr == nullptr || nnz_out == nullptr) return fail(IHG_ERR_INVALID, "ihg_merge_id_lists: bad argument");
    const int64_t nnz = n_rows > 0 ? ptr[n_rows] : 0;
    if (nnz > 0 && ids == nullptr) return fail(IHG_ERR_INVALID, "ihg_merge_id_lists: null buffer");
    // every row sorted on its own (rows are short; the few long ones of a power-law graph sort in place too), rows dealt to threads in contiguous ranges of about
    // equal entry counts; then distinct ids are counted per row, the counts prefix-summed, and the (id, multiplicity) pairs written - ascending id inside a row.
    // An entry is the 64-bit key (id << 32 | bits of its weight): one sort orders the ids and carries the weights (positive floats order like their bit patterns;
    // the order of equal ids only fixes the order their weights are added in - exact for the integer weights a multiplicity is).
    // out_ids == nullptr: COUNT ONLY (out_ptr and *nnz_out are written): what a caller needs to decide whether the merged list is worth building.
    std::vector<uint64_t> sorted(static_cast<size_t>(nnz));
    const int n_threads = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>({static_cast<int64_t>(std::thread::hardware_concurrency()), int64_t{32}, nnz / 200000 + 1})));
    std::vector<int64_t> cut(static_cast<size_t>(n_threads) + 1, n_rows);
    cut[0] = 0;
    for (int t = 1; t < n_threads; ++t)
        cut[t] = std::upper_bound(ptr, ptr + n_rows + 1, static_cast<int32_t>(nnz * t / n_threads)) - ptr - 1;
    auto over_ranges = [&](auto&& body) {
        std::vector<std::thread> pool;
        for (int t = 1; t < n_threads; ++t) pool.emplace_back([&, t] { body(cut[t], cut[t + 1]); });
        body(cut[0], cut[1]);
        for (auto& th : pool) th.join();
    };
    out_ptr[0] = 0;
    over_ranges([&](int64_t r0, int64_t r1) {
        for (int64_t k = ptr[r0]; k < ptr[r1]; ++k) {
            uint32_t wbits = 0x3f800000u;                    // 1.0f
            if (weights != nullptr) std::memcpy(&wbits, weights + k, sizeof(wbits));
            sorted[static_cast<size_t>(k)] = (static_cast<uint64_t>(static_cast<uint32_t>(ids[k])) << 32) | wbits;
        }
        for (int64_t r = r0; r < r1; ++r) {
            uint64_t* b = sorted.data() + ptr[r];
            uint64_t* e = sorted.data() + ptr[r + 1];
            std::sort(b, e);
            int32_t distinct = 0;
            for (uint64_t* p = b; p < e; ++p) distinct += (p == b || (p[0] >> 32) != (p[-1] >> 32)) ? 1 : 0;
            out_ptr[r + 1] = distinct;
        }
    });
    for (int64_t r = 0; r < n_rows; ++r) out_ptr[r + 1] += out_ptr[r];
    *nnz_out = n_rows > 0 ? out_ptr[n_rows] : 0;
    if (out_ids == nullptr) return IHG_OK;
    if (nnz > 0 && out_counts == nullptr) return fail(IHG_ERR_INVALID, "ihg_merge_id_lists: null buffer");
    over_ranges([&](int64_t r0, int64_t r1) {
        for (int64_t r = r0; r < r1; ++r) {
            const uint64_t* b = sorted.data() + ptr[r];
            const uint64_t* e = sorted.data() + ptr[r + 1];
            int64_t k = out_ptr[r];
            for (const uint64_t* p = b; p < e;) {
                const uint64_t* q = p;
                float total = 0.f;
                while (q < e && (*q >> 32) == (*p >> 32)) {
                    const uint32_t wbits = static_cast<uint32_t>(*q);
                    float w;
                    std::memcpy(&w, &wbits, sizeof(w));
                    total += w;
                    ++q;
                }
                out_ids[k] = static_cast<int32_t>(*p >> 32);
                out_counts[k] = total;
                ++k;
                p = q;
            }
        }
    });
    return IHG_OK;
}

int ihg_unique_triples(const int64_t* triples, int64_t n_edges, int64_t n_users, int64_t n_queries, int64_t n_items, int64_t* unique_out, float* count_out,
                       int32_t* file_to_unique, int64_t* n_unique_out) {
    if (n_edges < 0 || n_users < 0 || n_queries < 0 || n_items < 0 || n_unique_out == nullptr) return fail(IHG_ERR_INVALID, "ihg_unique_triples: bad argument");
    if (n_edges >= INT32_MAX) return fail(IHG_ERR_INVALID, "ihg_unique_triples: more than 2^31 interactions");
    *n_unique_out = 0;
    if (n_edges == 0) return IHG_OK;
    if (triples == nullptr) return fail(IHG_ERR_INVALID, "ihg_unique_triples: null buffer");
    const int64_t limit[3] = {n_users, n_queries, n_items};
    // bucket by user (a counting sort that keeps file order inside a user), then every user's run sorted by (query, item, file position) on the threads;
    // equal (query, item) neighbours of a run are one distinct hyperedge
    std::vector<int64_t> uptr(static_cast<size_t>(n_users) + 1, 0);
    for (int64_t e = 0; e < n_edges; ++e) {
        for (int m = 0; m < 3; ++m) {
            const int64_t local = triples[e * 3 + m];
            if (local < 0 || local >= limit[m])
                return fail(IHG_ERR_INVALID, "ihg_unique_triples: interaction %lld member %d id %lld out of range [0,%lld)", static_cast<long long>(e), m,
                            static_cast<long long>(local), static_cast<long long>(limit[m]));
        }
        ++uptr[static_cast<size_t>(triples[e * 3]) + 1];
    }
    for (int64_t u = 0; u < n_users; ++u) uptr[u + 1] += uptr[u];
    struct Entry {
        int64_t key;        // query * n_items + item
        int32_t file;       // position in the caller's array
        bool operator<(const Entry& o) const { return key != o.key ? key < o.key : file < o.file; }
    };
    std::vector<Entry> entries(static_cast<size_t>(n_edges));
    {
        std::vector<int64_t> cursor(uptr.begin(), uptr.end() - 1);
        for (int64_t e = 0; e < n_edges; ++e)
            entries[static_cast<size_t>(cursor[triples[e * 3]]++)] = Entry{triples[e * 3 + 1] * n_items + triples[e * 3 + 2], static_cast<int32_t>(e)};
    }
    const int n_threads = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>({static_cast<int64_t>(std::thread::hardware_concurrency()), int64_t{32}, n_edges / 200000 + 1})));
    std::vector<int64_t> cut(static_cast<size_t>(n_threads) + 1, n_users);
    cut[0] = 0;
    for (int t = 1; t < n_threads; ++t)
        cut[t] = std::upper_bound(uptr.begin(), uptr.end(), n_edges * t / n_threads) - uptr.begin() - 1;
    auto over_ranges = [&](auto&& body) {
        std::vector<std::thread> pool;
        for (int t = 1; t < n_threads; ++t) pool.emplace_back([&, t] { body(cut[t], cut[t + 1]); });
        body(cut[0], cut[1]);
        for (auto& th : pool) th.join();
    };
    std::vector<int64_t> first(static_cast<size_t>(n_users) + 1, 0);       // distinct hyperedges of the users before u
    over_ranges([&](int64_t u0, int64_t u1) {
        for (int64_t u = u0; u < u1; ++u) {
            Entry* b = entries.data() + uptr[u];
            Entry* e = entries.data() + uptr[u + 1];
            std::sort(b, e);
            int64_t distinct = 0;
            for (Entry* p = b; p < e; ++p) distinct += (p == b || p[0].key != p[-1].key) ? 1 : 0;
            first[u + 1] = distinct;
        }
    });
    for (int64_t u = 0; u < n_users; ++u) first[u + 1] += first[u];
    *n_unique_out = first[n_users];
    if (unique_out == nullptr) return IHG_OK;                                // count only
    if (count_out == nullptr) return fail(IHG_ERR_INVALID, "ihg_unique_triples: null buffer");
    over_ranges([&](int64_t u0, int64_t u1) {
        for (int64_t u = u0; u < u1; ++u) {
            const Entry* b = entries.data() + uptr[u];
            const Entry* e = entries.data() + uptr[u + 1];
            int64_t k = first[u];
            for (const Entry* p = b; p < e;) {
                const Entry* q = p;
                while (q < e && q->key == p->key) {
                    if (file_to_unique != nullptr) file_to_unique[q->file] = static_cast<int32_t>(k);
                    ++q;
                }
                unique_out[k * 3] = u;
                unique_out[k * 3 + 1] = p->key / n_items;
                unique_out[k * 3 + 2] = p->key % n_items;
                count_out[k] = static_cast<float>(q - p);
                ++k;
                p = q;
            }
        }
    });
    return IHG_OK;
}
}  // extern "C"
