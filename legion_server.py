#!/usr/bin/env python3
"""Entry point with the reference's name and CLI (legion_server.py:116-125):
python legion_server.py --dataset_path 'dataset' --dataset_name ukunion --train_batch_size 8000
    --fanout [25,10] --gpu_number 2 --epoch 2 --cache_memory 38000000"""
import sys

from legion_amd.launcher import main

if __name__ == "__main__":
    sys.exit(main())
