"""ORACLE (test infrastructure): state_dict key names + shapes of the reference UNet for a config.

Restates the constructor wiring of models/unet_2d_condition.py:160-568 and the five block types
of models/unet_2d_blocks.py used by SD-v1.5 / SDXL. Checked against the reference's own
`state_dict()` in tests/test_oracle_vs_reference.py and against the engine's weight table
(rt_weight_info) in tests/test_abi.py.
"""
from collections import OrderedDict


def weight_shapes(cfg):
    s = OrderedDict()
    boc = cfg["block_out_channels"]
    nlev = len(boc)
    temb = boc[0] * 4
    ctxd = cfg["cross_attention_dim"]
    lin_proj = cfg["use_linear_projection"]

    def tup(v):
        return tuple(v) if isinstance(v, (tuple, list)) else (v,) * nlev
    lpb, tl, heads = tup(cfg["layers_per_block"]), tup(cfg["transformer_layers_per_block"]), tup(cfg["attention_head_dim"])

    def lin(name, i, o, bias=True):
        s[name + ".weight"] = (o, i)
        if bias:
            s[name + ".bias"] = (o,)

    def conv(name, i, o, k):
        s[name + ".weight"] = (o, i, k, k)
        s[name + ".bias"] = (o,)

    def norm(name, c):
        s[name + ".weight"] = (c,)
        s[name + ".bias"] = (c,)

    def resnet(name, cin, cout):
        norm(name + ".norm1", cin)
        conv(name + ".conv1", cin, cout, 3)
        lin(name + ".time_emb_proj", temb, cout)
        norm(name + ".norm2", cout)
        conv(name + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(name + ".conv_shortcut", cin, cout, 1)

    def transformer(name, c, nl):
        norm(name + ".norm", c)
        if lin_proj:
            lin(name + ".proj_in", c, c)
        else:
            conv(name + ".proj_in", c, c, 1)
        for li in range(nl):
            b = f"{name}.transformer_blocks.{li}"
            norm(b + ".norm1", c)
            lin(b + ".attn1.to_q", c, c, False); lin(b + ".attn1.to_k", c, c, False); lin(b + ".attn1.to_v", c, c, False)
            lin(b + ".attn1.to_out.0", c, c)
            norm(b + ".norm2", c)
            lin(b + ".attn2.to_q", c, c, False); lin(b + ".attn2.to_k", ctxd, c, False); lin(b + ".attn2.to_v", ctxd, c, False)
            lin(b + ".attn2.to_out.0", c, c)
            norm(b + ".norm3", c)
            lin(b + ".ff.net.0.proj", c, c * 8)
            lin(b + ".ff.net.2", c * 4, c)
        if lin_proj:
            lin(name + ".proj_out", c, c)
        else:
            conv(name + ".proj_out", c, c, 1)

    conv("conv_in", cfg["in_channels"], boc[0], 3)
    lin("time_embedding.linear_1", boc[0], temb)
    lin("time_embedding.linear_2", temb, temb)
    if cfg.get("addition_embed_type") == "text_time":
        lin("add_embedding.linear_1", cfg["projection_class_embeddings_input_dim"], temb)
        lin("add_embedding.linear_2", temb, temb)
    out_c = boc[0]
    for i, bt in enumerate(cfg["down_block_types"]):
        in_c, out_c = out_c, boc[i]
        for j in range(lpb[i]):
            resnet(f"down_blocks.{i}.resnets.{j}", in_c if j == 0 else out_c, out_c)
        if bt == "CrossAttnDownBlock2D":
            for j in range(lpb[i]):
                transformer(f"down_blocks.{i}.attentions.{j}", out_c, tl[i])
        if i != nlev - 1:
            conv(f"down_blocks.{i}.downsamplers.0.conv", out_c, out_c, 3)
    resnet("mid_block.resnets.0", boc[-1], boc[-1])
    transformer("mid_block.attentions.0", boc[-1], tl[-1])
    resnet("mid_block.resnets.1", boc[-1], boc[-1])
    rboc = list(reversed(boc))
    rlpb, rtl = list(reversed(lpb)), list(reversed(tl))
    out_c = rboc[0]
    for i, bt in enumerate(cfg["up_block_types"]):
        prev = out_c
        out_c = rboc[i]
        in_c = rboc[min(i + 1, nlev - 1)]
        nl = rlpb[i] + 1
        for j in range(nl):
            skip_c = in_c if j == nl - 1 else out_c
            res_in = prev if j == 0 else out_c
            resnet(f"up_blocks.{i}.resnets.{j}", res_in + skip_c, out_c)
        if bt == "CrossAttnUpBlock2D":
            for j in range(nl):
                transformer(f"up_blocks.{i}.attentions.{j}", out_c, rtl[i])
        if i != nlev - 1:
            conv(f"up_blocks.{i}.upsamplers.0.conv", out_c, out_c, 3)
    norm("conv_norm_out", boc[0])
    conv("conv_out", boc[0], cfg["out_channels"], 3)
    return s
