import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/examples")
import torch
import dlrm_dcn_v2 as ex
import keras_rs_amd.layers as kl
dev = torch.device("cuda", 0)
hots = [3, 1, 2]
m = ex.build_model(64, 500, hots, embedding_dim=32, projection=16, cross_layers=1, bottom=(64, 32), top=(8, 1))
x, y = ex.synthetic_batch(64, 13, 500, hots, dev)
d = m.bottom_mlp(x["dense_input"])
e = m.embedding_layer(x["large_emb_inputs"])
vals = list(e.values())
print("dense", d.dtype, d.shape, "emb", vals[0].dtype, vals[0].shape, "slab info", [getattr(v, "_krs_slab", None) and (v._krs_slab[0].shape, v._krs_slab[1:]) for v in vals])
x0 = kl.concat_features([d.to(vals[0].dtype), *vals])
print("x0 is slab:", x0.data_ptr() == vals[0]._krs_slab[0].data_ptr(), x0.shape)
