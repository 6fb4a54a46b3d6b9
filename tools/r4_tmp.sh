export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -s -k "without_hyperedge_rows or layer0_path or f8_ or heaviest_rows" 2>&1 | grep -E "hyperedges:|passed|failed|Error|assert" | tail -14
for v in 1 0; do echo "grouped=$v"; IHG_NODE_FWD_GROUPED=$v python3 tools/kbench.py --config C3 --rounds 8 --ops layer 2>/dev/null | grep -E "node_interact_fwd"; done
for v in 1 0; do echo "C4 grouped=$v"; IHG_NODE_FWD_GROUPED=$v python3 tools/kbench.py --config C4 --rounds 6 --ops layer 2>/dev/null | grep -E "node_interact_fwd"; done
