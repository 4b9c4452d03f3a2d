import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
if mode == "ours_first":
    import numpy as np
    import sfmlocalization_amd as S
    from sfmlocalization_amd import synth
    m = synth.make_map(1, 4, desc_per_view=128)
    dm = S.Map(m.view_id, m.view_off, m.desc)
    print("ours ok", S.device_count(), flush=True)
    import torch
    x = torch.zeros(4, device="cuda")
    print("torch ok", flush=True)
else:
    import torch
    x = torch.zeros(4, device="cuda")
    print("torch ok", flush=True)
    import sfmlocalization_amd as S
    from sfmlocalization_amd import synth
    m = synth.make_map(1, 4, desc_per_view=128)
    dm = S.Map(m.view_id, m.view_off, m.desc)
    print("ours ok", flush=True)
