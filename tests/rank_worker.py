"""A rank of the launcher test: joins the group torch.distributed.run set up (gloo), all-reduces
its rank and prints one JSON line from rank 0 -- the shape of what bench.py does under the launcher."""
import json
import os
import sys

import torch
import torch.distributed as dist

dist.init_process_group('gloo')
t = torch.tensor([float(dist.get_rank() + 1)])
dist.all_reduce(t)
dist.barrier()
if dist.get_rank() == 0:
    print(json.dumps({'world': dist.get_world_size(), 'env_world': int(os.environ['WORLD_SIZE']), 'sum': float(t[0]),
                      'argv': sys.argv[1:], 'master': os.environ.get('MASTER_ADDR')}), flush=True)
dist.destroy_process_group()
