#!/bin/bash
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --backend gloo --share-gpu --seqs 256 --steps 20 --warmup 4 --repeats 1 --no-extras --no-cpu-baseline 2>/dev/null | cut -c1-400
timeout -k 10 300 python -m pytest tests/test_facade_gpu.py tests/test_wire.py tests/test_replay.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
