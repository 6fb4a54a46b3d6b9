for name in "$@"; do echo $name; IHGNN_HIP_LIBRARY=/root/repo/build_ab/lib_$name.so python3 /root/repo/tools/kbench.py --config C3 --rounds 6 --ops linear 2>&1 | grep node_linear; done
