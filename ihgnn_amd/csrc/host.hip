// host.cpp (compiled by hipcc with the rest) - host-side entry points: ABI version, error string, hypergraph / pairwise-graph
// layout builders, CSR transpose, search-log CSV ingestion.
#include "common.hpp"

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

int32_t ihg_abi_version(void) { return 21; }

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
                          int64_t* neg, int64_t neg_capacity) {
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
}  // extern "C"
