#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/graph_node_order_probe.log; rm -f $L
for n in 300 3000; do timeout -k 10 300 python tools/graph_node_order_probe.py $n 5 >> $L 2>gpurun_out/graph_node_order_probe.err; echo "exit $?" >> $L; done
tail -60 $L
