"""Token-map producer: drop-in for `utils/attention_utils.py:get_token_maps`.

Same arguments and return value as the reference (list of [1,4,h,w] region masks on the GPU).  The figures the reference writes
to `save_dir` on EVERY call (attention_utils.py:266-270,334-335) are produced only when `return_vis=True` asks for them: the call then
returns the reference's triple (masks, segments_vis, token_maps_vis) - uint8 RGB renderings of the cluster map and of the per-span
maps (matplotlib only; seaborn, which the reference uses for the heat maps, is not a dependency).
The numerical steps follow attention_utils.py:233-341 line by line: 32x32 self-attention affinity ->
SpectralClustering(n_init=100, kmeans) -> clusters labelled by min-max-normalised cross-attention score against
`segment_threshold` -> bicubic(antialias) resize, clamp, normalise.  The clustering itself stays on the CPU
(scikit-learn, seeded like the reference: `seed_everything(seed)` before `fit_predict`)."""
import os
import random

import numpy as np
import torch

# module-name lists of the reference (attention_utils.py:12-67): data, not code
SelfAttentionLayers = [
    'down_blocks.0.attentions.0.transformer_blocks.0.attn1', 'down_blocks.0.attentions.1.transformer_blocks.0.attn1',
    'down_blocks.1.attentions.0.transformer_blocks.0.attn1', 'down_blocks.1.attentions.1.transformer_blocks.0.attn1',
    'down_blocks.2.attentions.0.transformer_blocks.0.attn1', 'down_blocks.2.attentions.1.transformer_blocks.0.attn1',
    'mid_block.attentions.0.transformer_blocks.0.attn1',
    'up_blocks.1.attentions.0.transformer_blocks.0.attn1', 'up_blocks.1.attentions.1.transformer_blocks.0.attn1',
    'up_blocks.1.attentions.2.transformer_blocks.0.attn1', 'up_blocks.2.attentions.0.transformer_blocks.0.attn1',
    'up_blocks.2.attentions.1.transformer_blocks.0.attn1', 'up_blocks.2.attentions.2.transformer_blocks.0.attn1',
    'up_blocks.3.attentions.0.transformer_blocks.0.attn1', 'up_blocks.3.attentions.1.transformer_blocks.0.attn1',
    'up_blocks.3.attentions.2.transformer_blocks.0.attn1',
]
CrossAttentionLayers = [
    'down_blocks.1.attentions.0.transformer_blocks.0.attn2', 'down_blocks.2.attentions.0.transformer_blocks.0.attn2',
    'down_blocks.2.attentions.1.transformer_blocks.0.attn2', 'mid_block.attentions.0.transformer_blocks.0.attn2',
    'up_blocks.1.attentions.0.transformer_blocks.0.attn2', 'up_blocks.1.attentions.1.transformer_blocks.0.attn2',
    'up_blocks.1.attentions.2.transformer_blocks.0.attn2', 'up_blocks.2.attentions.1.transformer_blocks.0.attn2',
]
CrossAttentionLayers_XL = [
    'down_blocks.2.attentions.1.transformer_blocks.3.attn2', 'down_blocks.2.attentions.1.transformer_blocks.4.attn2',
    'mid_block.attentions.0.transformer_blocks.0.attn2', 'mid_block.attentions.0.transformer_blocks.1.attn2',
    'mid_block.attentions.0.transformer_blocks.2.attn2', 'mid_block.attentions.0.transformer_blocks.3.attn2',
    'up_blocks.0.attentions.0.transformer_blocks.1.attn2', 'up_blocks.0.attentions.0.transformer_blocks.2.attn2',
    'up_blocks.0.attentions.0.transformer_blocks.3.attn2', 'up_blocks.0.attentions.0.transformer_blocks.4.attn2',
    'up_blocks.0.attentions.0.transformer_blocks.5.attn2', 'up_blocks.0.attentions.0.transformer_blocks.6.attn2',
    'up_blocks.0.attentions.0.transformer_blocks.7.attn2', 'up_blocks.1.attentions.0.transformer_blocks.0.attn2',
]


def seed_everything(seed):                         # utils/richtext_utils.py:22-27
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def _segment(selfattn_maps, crossattn_maps, seed, num_segments, resolution):
    """attention_utils.py:243-283: the part of get_token_maps that depends only on the recorded maps, the seed and the segment count -
    the 32x32 affinity, its spectral clustering and the averaged cross-attention maps."""
    from sklearn.cluster import SpectralClustering
    maps32 = []
    for attn_map in selfattn_maps.values():
        res_map = int(np.sqrt(attn_map.shape[1]))
        if res_map != resolution:
            continue
        # attention_utils.py:246-251 resizes every kept map to (resolution, resolution) - but only maps that already ARE 32 x 32 get
        # here, and a same-size bicubic(antialias) resize returns its input bit for bit (weights 1 / 0; pinned by
        # tests/test_token_maps.py::test_same_size_bicubic_antialias_resize_is_the_identity): 60 resizes of 1024 x 1024 skipped
        maps32.append(attn_map.reshape(1, resolution ** 2, res_map ** 2).float())
    # averaged where the maps live: the facades hand over GPU tensors (240 MB at SDXL) and only the 1024 x 1024 result crosses to the host
    # (the reference moves every map to the CPU first, attention_utils.py:246; fp32 mean either way)
    affinity = torch.cat(maps32).mean(0).cpu().numpy()
    seed_everything(seed)
    sc = SpectralClustering(num_segments, affinity='precomputed', n_init=100, assign_labels='kmeans')
    # 100 k-means restarts on 1024 points x num_segments coordinates: every OpenMP / BLAS region is far too small for the host's
    # thread pool (256 hardware threads on the GPU boxes) and pays its fork / join instead - one thread is the fastest setting
    # (0.98 -> 0.59 s on 8 cores).  Same seed, same labels (tests/test_token_maps.py); RTDIFF_CLUSTER_THREADS=0 leaves the pools alone.
    nthr = int(os.environ.get("RTDIFF_CLUSTER_THREADS", "1"))
    if nthr > 0:
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=nthr):
            clusters = sc.fit_predict(affinity)
    else:
        clusters = sc.fit_predict(affinity)
    clusters = clusters.reshape(resolution, resolution)

    cross = []
    for attn_map in crossattn_maps.values():
        res_map = int(np.sqrt(attn_map.shape[1]))
        a = attn_map.reshape(1, res_map, res_map, -1).permute([0, 3, 1, 2]).float().cpu()
        a = torch.nn.functional.interpolate(a, (resolution, resolution), mode='bicubic', antialias=True)
        cross.append(a.permute([0, 2, 3, 1]))
    cross = torch.cat(cross).mean(0).cpu().numpy()
    return clusters, cross


def get_token_maps(selfattn_maps, crossattn_maps, n_maps, save_dir, width, height, obj_tokens, seed=0, tokens_vis=None,
                   preprocess=False, segment_threshold=0.3, num_segments=5, return_vis=False, save_attn=False, device=None, cache=None):
    """`cache` (not in the reference's signature; optional): a dict the caller keeps across calls ON THE SAME RECORDED MAPS.  The
    reference calls this function twice per image (sample.py:78-95: colour-object masks, then region masks) and both calls redo the
    same seeded clustering of the same affinity - 60 maps of 1024 x 1024 averaged on the host plus SpectralClustering(n_init=100),
    0.9 s of a 10 s SDXL image.  With a cache the second call reuses `clusters` and the averaged cross-attention maps of the first:
    same seed, same inputs, same labels (tests/test_token_maps.py)."""
    resolution = 32
    key = (seed, num_segments, resolution)
    if cache is not None and cache.get("key") == key:
        clusters, cross = cache["clusters"], cache["cross"]
        seed_everything(seed)                      # (nothing below draws random numbers; sample.py reseeds before every stage)
    else:
        clusters, cross = _segment(selfattn_maps, crossattn_maps, seed, num_segments, resolution)
        if cache is not None:
            cache.update(key=key, clusters=clusters, cross=cross)
    normalized_span_maps = []
    for token_ids in obj_tokens:
        span = cross[:, :, token_ids.numpy()]
        nm = np.zeros_like(span)
        for i in range(span.shape[-1]):
            cur = span[:, :, i]
            nm[:, :, i] = (cur - np.abs(cur.min())) / (cur.max() - cur.min())
        normalized_span_maps.append(nm)
    foreground = [np.zeros([clusters.shape[0], clusters.shape[1]]).squeeze() for _ in normalized_span_maps]
    background = np.zeros([clusters.shape[0], clusters.shape[1]]).squeeze()
    for c in range(num_segments):
        cluster_mask = np.zeros_like(clusters)
        cluster_mask[clusters == c] = 1.
        is_fg = False
        for nm, fg, token_ids in zip(normalized_span_maps, foreground, obj_tokens):
            scores = [(cluster_mask * nm[:, :, i]).sum() / cluster_mask.sum() for i in range(len(token_ids))]
            if max(scores) > segment_threshold:
                fg += cluster_mask
                is_fg = True
        if not is_fg:
            background += cluster_mask
    foreground.append(background)
    resized = torch.cat([torch.nn.functional.interpolate(torch.from_numpy(m).unsqueeze(0).unsqueeze(0), (height, width),
                                                         mode='bicubic', antialias=True)[0] for m in foreground]).clamp(0, 1)
    resized = resized / (resized.sum(0, True) + 1e-8)
    dev = device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu")
    dtype = next(iter(crossattn_maps.values())).dtype
    maps = [m.unsqueeze(0).unsqueeze(1).repeat([1, 4, 1, 1]).to(dtype).to(dev) for m in resized]
    if not return_vis:
        return maps
    # attention_utils.py:263-277,334-339: the Gradio apps show these two pictures next to the image
    segments_vis = _render_clusters(clusters, save_dir, num_segments, seed)
    token_maps_vis = _render_token_maps([[m[None] for m in foreground], [m[None] for m in resized.numpy()]], obj_tokens, save_dir, seed, tokens_vis)
    return maps, segments_vis, token_maps_vis


def _canvas_rgb(fig):
    fig.canvas.draw()
    w, h = fig.canvas.get_width_height()
    return np.asarray(fig.canvas.buffer_rgba(), dtype=np.uint8).reshape(h, w, 4)[:, :, :3].copy()


def _render_clusters(clusters, save_dir, num_segments, seed):
    """attention_utils.py:266-275: imshow of the 32 x 32 label map, axes off; uint8 RGB [H, W, 3]."""
    import matplotlib
    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt
    fig = plt.figure()
    plt.imshow(clusters)
    plt.axis('off')
    if save_dir:
        plt.savefig(os.path.join(save_dir, 'segmentation_k%d_seed%d.jpg' % (num_segments, seed)), bbox_inches='tight', pad_inches=0)
    vis = _canvas_rgb(fig)
    plt.close(fig)
    return vis


def _render_token_maps(atten_map_list, obj_tokens, save_dir, seed, tokens_vis=None):
    """attention_utils.py:96-146 (plot_attention_maps): one row of heat maps per list entry - the cluster-level span maps and the
    resized, normalised region masks - 'OrRd' colour map on a common scale, span tokens as titles, the last map titled 'other tokens';
    like the reference, returns the rendering of the LAST row (uint8 RGB [H, W, 3]) and writes average_seed<seed>_attn<i>.png."""
    import matplotlib
    matplotlib.use("Agg", force=False)
    import matplotlib as mpl
    import matplotlib.pyplot as plt
    img = None
    for i, attn_map in enumerate(atten_map_list):
        n_obj = len(attn_map)
        fig, axs = plt.subplots(ncols=n_obj + 1, gridspec_kw=dict(width_ratios=[1] * n_obj + [0.1]))
        fig.set_figheight(3)
        fig.set_figwidth(3 * n_obj + 0.1)
        cmap = plt.get_cmap('OrRd')
        vmax = max([0.0] + [float(np.asarray(m).max()) for m in attn_map])
        vmin = min([1.0] + [float(np.asarray(m).min()) for m in attn_map])
        for tid in range(n_obj):
            axs[tid].imshow(np.asarray(attn_map[tid][0]), cmap=cmap, vmin=vmin, vmax=vmax, aspect='auto', interpolation='nearest')
            axs[tid].set_axis_off()
            if tokens_vis is not None:
                if tid == n_obj - 1:
                    label = 'other tokens'
                else:
                    label = ''.join(' ' + tokens_vis[int(t) - 1][:-len('</w>')] for t in obj_tokens[tid])
                axs[tid].set_title(label)
        fig.colorbar(mpl.cm.ScalarMappable(norm=mpl.colors.Normalize(vmin=vmin, vmax=vmax), cmap=cmap), cax=axs[-1])
        if save_dir:
            fig.savefig(os.path.join(save_dir, 'average_seed%d_attn%d.png' % (seed, i)), dpi=100)
        img = _canvas_rgb(fig)
        plt.close(fig)
    return img
