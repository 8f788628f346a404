"""TEST INFRASTRUCTURE: expected outputs of the REFERENCE rich-text front-end (utils/richtext_utils.py) on the example
JSONs of the reference README / gradio apps, with a stub tokenizer (no CLIP vocabulary offline; transformers 5.x has no
`_tokenize`).  Writes tests/golden/richtext_cases.json.   python -m oracle.make_richtext_golden"""
import importlib
import json
import os
import re

import torch

from .refload import load_reference

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "richtext_cases.json")


class StubTokenizer:
    """Whitespace/punctuation word splitter with CLIP-style '</w>' suffixes."""

    def _tokenize(self, text):
        return [w + '</w>' for w in re.findall(r"[a-z0-9]+|[^\sa-z0-9]", text.lower())]


CASES = [
    # README.md:63,73,83,93 style examples (colour / footnote / style / size) + mixed ones
    {"ops": [{"insert": "a Gothic "}, {"attributes": {"color": "#b26b00"}, "insert": "church"}, {"insert": " in a the sunset with a beautiful landscape in the background.\n"}]},
    {"ops": [{"insert": "A mesmerizing sight that captures the beauties of this "}, {"attributes": {"link": "A charming wooden house nestled among the trees."}, "insert": "cabin"},
             {"insert": " in the woods near a "}, {"attributes": {"link": "A serene lake with ducks."}, "insert": "lake"}, {"insert": ".\n"}]},
    {"ops": [{"insert": "a "}, {"attributes": {"font": "mirza"}, "insert": "beautiful garden"}, {"insert": " with a "},
             {"attributes": {"font": "roboto"}, "insert": "snow mountain in the background"}, {"insert": "\n"}]},
    {"ops": [{"insert": "A pizza with "}, {"attributes": {"size": "50px"}, "insert": "pineapples"}, {"insert": ", pepperonis, and mushrooms on the top\n"}]},
    {"ops": [{"insert": "a "}, {"attributes": {"font": "slabo"}, "insert": "night sky"}, {"insert": " "}, {"attributes": {"font": "slabo"}, "insert": "filled with stars"},
             {"insert": " above a "}, {"attributes": {"color": "#ff0000", "size": "30px", "strike": True}, "insert": "red"}, {"insert": " "},
             {"attributes": {"color": "#ff0000"}, "insert": "barn"}, {"insert": " and a "}, {"attributes": {"size": "18px", "strike": True}, "insert": "fence"},
             {"insert": " near a barn\n"}]},
]


def main():
    mods = load_reference()
    ru = importlib.import_module("utils.richtext_utils")
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    model = type("M", (), {"tokenizer": StubTokenizer()})()
    out = []
    try:
        for js in CASES:
            base, styles, notes, note_tok, ctext, cnames, crgbs, sizes, use_grad = ru.parse_json(js)
            prompts, ids, base_tokens = ru.get_region_diffusion_input(model, base, styles, notes, note_tok, ctext, cnames)
            tfd = ru.get_attention_control_input(model, base_tokens, sizes)
            tfd, cids = ru.get_gradient_guidance_input(model, base_tokens, ctext, crgbs, tfd, color_guidance_weight=0.5)
            out.append({"json": js, "base": base, "styles": styles, "notes": notes, "note_tokens": note_tok, "color_text": ctext,
                        "color_names": cnames, "color_rgbs": [c.flatten().tolist() for c in crgbs], "sizes": sizes, "use_grad": use_grad,
                        "region_prompts": prompts, "region_ids": [i.tolist() for i in ids], "base_tokens": base_tokens,
                        "word_pos": None if tfd["word_pos"] is None else tfd["word_pos"].tolist(),
                        "font_size": None if tfd["font_size"] is None else tfd["font_size"].tolist(),
                        "color_ids": [i.tolist() for i in cids], "guidance_start_step": tfd["guidance_start_step"],
                        "color_guidance_weight": tfd["color_guidance_weight"]})
    finally:
        torch.Tensor.cuda = orig
    json.dump(out, open(OUT, "w"), indent=1)
    print("wrote", OUT, len(out), "cases")


if __name__ == "__main__":
    main()
