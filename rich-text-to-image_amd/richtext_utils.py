"""Rich-text (Quill delta JSON) front-end: drop-in for `utils/richtext_utils.py`.

Function names, argument order and return tuples are the reference's (utils/richtext_utils.py:74-234) so `sample.py`
runs unchanged; the implementation is a fresh one around a small span parser.  Reference quirks that are part of the
observable behaviour are kept and marked.  Tensors go to `device` (default: CUDA when
available) where the reference hard-codes `.cuda()`.
"""
import os
import random
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

# named colours used to turn a hex colour into a word of the region prompt (richtext_utils.py:7-19; data)
COLORS = {'brown': [165, 42, 42], 'red': [255, 0, 0], 'pink': [253, 108, 158], 'orange': [255, 165, 0], 'yellow': [255, 255, 0],
          'purple': [128, 0, 128], 'green': [0, 128, 0], 'blue': [0, 0, 255], 'white': [255, 255, 255], 'gray': [128, 128, 128],
          'black': [0, 0, 0]}
# editor font -> artistic style (richtext_utils.py:59-71; data, the font whitelist of the HTML editor)
FONT_STYLES = {'mirza': 'Claud Monet, impressionism, oil on canvas', 'roboto': 'Ukiyoe',
               'cursive': 'Cyber Punk, futuristic, blade runner, william gibson, trending on artstation hq',
               'sofia': 'Pop Art, masterpiece, andy warhol', 'slabo': 'Vincent Van Gogh', 'inconsolata': 'Pixel Art, 8 bits, 16 bits',
               'ubuntu': 'Rembrandt', 'Monoton': 'neon art, colorful light, highly details, octane render',
               'Akronim': 'Abstract Cubism, Pablo Picasso'}


def _device(device=None):
    return device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu")


def seed_everything(seed):
    random.seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


def find_nearest_color(rgb):
    if isinstance(rgb, (list, tuple)):
        rgb = torch.FloatTensor(rgb)[None, :, None, None] / 255.
    rgb = rgb.detach().cpu()
    names = list(COLORS)
    dist = [float(np.linalg.norm(rgb - torch.FloatTensor(COLORS[n])[None, :, None, None] / 255.)) for n in names]
    return names[int(np.argmin(dist))]


def hex_to_rgb(hex_string, return_nearest_color=False, device=None):
    h = hex_string.lstrip('#')
    rgb = torch.FloatTensor([int(h[i:i + 2], 16) for i in (0, 2, 4)])[None, :, None, None] / 255.
    if return_nearest_color:
        return rgb.to(_device(device)), find_nearest_color(rgb)
    return rgb.to(_device(device))


def font2style(font):
    return FONT_STYLES[font]


@dataclass
class RichText:
    base_text_prompt: str = ''
    style_text_prompts: List[str] = field(default_factory=list)
    footnote_text_prompts: List[str] = field(default_factory=list)
    footnote_target_tokens: List[str] = field(default_factory=list)
    color_text_prompts: List[str] = field(default_factory=list)
    color_names: List[str] = field(default_factory=list)
    color_rgbs: list = field(default_factory=list)
    size_text_prompts_and_sizes: list = field(default_factory=list)
    use_grad_guidance: bool = False

    def astuple(self):
        return (self.base_text_prompt, self.style_text_prompts, self.footnote_text_prompts, self.footnote_target_tokens,
                self.color_text_prompts, self.color_names, self.color_rgbs, self.size_text_prompts_and_sizes, self.use_grad_guidance)


def _span_font_size(attrs):
    """px/3; strike-through flips the sign (richtext_utils.py:113-120, quirk 6).  A strike without a size keeps 1."""
    if 'size' not in attrs:
        return 1
    size = float(attrs['size'][:-2]) / 3.
    return -size if 'strike' in attrs else size


def parse_rich_text(delta, device=None) -> RichText:
    out = RichText()
    last_style: Optional[str] = None
    for op in delta['ops']:
        text = op['insert'].rstrip('\n')
        out.base_text_prompt += text
        if text == ' ':
            continue                                  # a lone blank neither carries attributes nor resets the style run
        attrs = op.get('attributes')
        if attrs is None:
            continue                                  # (the reference also leaves `prev_style` untouched here)
        if 'font' in attrs:
            style = font2style(attrs['font'])
            if style == last_style:                   # adjacent spans of one font merge into one region prompt
                head = out.style_text_prompts[-1].split('in the style of')[0]
                out.style_text_prompts[-1] = head + ' ' + text + f' in the style of {style}'
            else:
                out.style_text_prompts.append(text + f' in the style of {style}')
            last_style = style
        else:
            last_style = None
        if 'link' in attrs:                           # footnote
            out.footnote_text_prompts.append(attrs['link'])
            out.footnote_target_tokens.append(text)
        if 'color' in attrs:
            out.use_grad_guidance = True
            rgb, name = hex_to_rgb(attrs['color'], True, device)
            # quirk 8: the reference never updates `prev_color_rgb`, so same-colour neighbours are never merged
            out.color_rgbs.append(rgb)
            out.color_names.append(name)
            out.color_text_prompts.append(text)
        fs = _span_font_size(attrs)
        if fs != 1:
            out.size_text_prompts_and_sizes.append([text, fs])
    return out


def parse_json(json_str, device=None):
    """Same 9-tuple as utils/richtext_utils.py:74-136."""
    return parse_rich_text(json_str, device).astuple()


def _token_ids(tokenizer, base_tokens, text):
    # quirk 7: `.index` maps a repeated word to its FIRST occurrence; +1 skips the start-of-text token
    return [base_tokens.index(tok) + 1 for tok in tokenizer._tokenize(text)]


def get_region_diffusion_input(model, base_text_prompt, style_text_prompts, footnote_text_prompts, footnote_target_tokens,
                               color_text_prompts, color_names):
    """Algorithm 1 of the paper (utils/richtext_utils.py:139-185): one region prompt per attributed span + the base prompt."""
    tok = model.tokenizer
    base_tokens = tok._tokenize(base_text_prompt)
    prompts, ids = [], []
    for p in style_text_prompts:
        prompts.append(p)
        ids.append(_token_ids(tok, base_tokens, p.split('in the style of')[0]))
    for note, target in zip(footnote_text_prompts, footnote_target_tokens):
        prompts.append(note)
        ids.append(_token_ids(tok, base_tokens, target))
    for text, name in zip(color_text_prompts, color_names):
        prompts.append(name + ' ' + text)
        ids.append(_token_ids(tok, base_tokens, text))
    prompts.append(base_text_prompt)
    used = {i for group in ids for i in group}
    ids.append([i for i in range(1, len(base_tokens) + 1) if i not in used])
    return prompts, [torch.LongTensor(g) for g in ids], base_tokens


def get_attention_control_input(model, base_tokens, size_text_prompts_and_sizes, device=None):
    word_pos, sizes = [], []
    for text, fs in size_text_prompts_and_sizes:
        for i in _token_ids(model.tokenizer, base_tokens, text):
            word_pos.append(i)
            sizes.append(fs)
    if not word_pos:
        return {'word_pos': None, 'font_size': None}
    dev = _device(device)
    return {'word_pos': torch.LongTensor(word_pos).to(dev), 'font_size': torch.FloatTensor(sizes).to(dev)}


def get_gradient_guidance_input(model, base_tokens, color_text_prompts, color_rgbs, text_format_dict, guidance_start_step=999,
                                color_guidance_weight=1):
    ids = [_token_ids(model.tokenizer, base_tokens, t) for t in color_text_prompts]
    used = {i for group in ids for i in group}
    ids.append([i for i in range(1, len(base_tokens) + 1) if i not in used])
    text_format_dict['target_RGB'] = color_rgbs
    text_format_dict['guidance_start_step'] = guidance_start_step
    text_format_dict['color_guidance_weight'] = color_guidance_weight
    return text_format_dict, [torch.LongTensor(g) for g in ids]
