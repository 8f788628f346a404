"""SURVEY 8f f3 (text side): CLIP BPE tokenizer on a synthetic vocabulary and the SDXL prompt-encoding rules (xl.py:318-440)
on tiny random CLIP encoders.  No vocabulary / checkpoint exists offline => published-algorithm checks only."""
import json

import pytest
import torch


@pytest.fixture(scope="module")
def tok(tmp_path_factory):
    from rich_text_to_image_amd import clip_tokenizer as ct
    d = tmp_path_factory.mktemp("vocab")
    alpha = list(ct._byte_alphabet().values())
    vocab = alpha + [a + "</w>" for a in alpha]
    merges = [("c", "a"), ("ca", "t</w>"), ("t", "h"), ("th", "e</w>"), ("d", "o"), ("do", "g</w>"), ("i", "n"), ("in", "g</w>"),
              ("r", "u"), ("ru", "n"), ("n", "ing</w>")]
    vocab += [a + b for a, b in merges] + ["<|startoftext|>", "<|endoftext|>"]
    json.dump({t: i for i, t in enumerate(vocab)}, open(d / "vocab.json", "w"))
    open(d / "merges.txt", "w").write("#version: 0.2\n" + "\n".join(a + " " + b for a, b in merges) + "\n")
    return ct.ClipBPETokenizer.from_pretrained(str(d))


def test_bpe_tokenize(tok):
    assert tok._tokenize("The cat's  running dog, 42 cats!") == ['the</w>', 'cat</w>', "'", 's</w>', 'run', 'ning</w>', 'dog</w>', ',</w>',
                                                                '4</w>', '2</w>', 'ca', 't', 's</w>', '!</w>']
    assert len(set(__import__("rich_text_to_image_amd.clip_tokenizer", fromlist=["x"])._byte_alphabet().values())) == 256
    ids = tok("the cat", padding="max_length", max_length=8, truncation=True, return_tensors="pt").input_ids
    assert ids.shape == (1, 8) and ids[0, 0] == tok.bos_token_id and ids[0, 3] == tok.eos_token_id and (ids[0, 4:] == tok.pad_token_id).all()
    long = tok("cat " * 20, padding="max_length", max_length=8, truncation=True).input_ids
    assert len(long) == 8 and long[-1] == tok.eos_token_id


def test_front_end_runs_on_bpe_tokens(tok):
    from rich_text_to_image_amd import richtext_utils as ru
    model = type("M", (), {"tokenizer": tok})()
    js = {"ops": [{"insert": "the "}, {"attributes": {"font": "slabo"}, "insert": "running cat"}, {"insert": " and the dog\n"}]}
    base, styles, *_ = ru.parse_json(js, device="cpu")
    prompts, ids, base_tokens = ru.get_region_diffusion_input(model, base, styles, [], [], [], [])
    assert base_tokens == ['the</w>', 'run', 'ning</w>', 'cat</w>', 'a', 'n', 'd</w>', 'the</w>', 'dog</w>']
    assert ids[0].tolist() == [base_tokens.index('run') + 1, base_tokens.index('ning</w>') + 1, base_tokens.index('cat</w>') + 1]
    assert sorted(ids[0].tolist() + ids[1].tolist()) == list(range(1, len(base_tokens) + 1))


def test_sdxl_prompt_encoding_rules(tok):
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
    from rich_text_to_image_amd.checkpoint import ClipEncodersXL
    torch.manual_seed(0)
    n_vocab = len(tok.encoder)
    c1 = CLIPTextConfig(vocab_size=n_vocab, hidden_size=32, intermediate_size=64, num_hidden_layers=3, num_attention_heads=2,
                        max_position_embeddings=77, eos_token_id=tok.eos_token_id, bos_token_id=tok.bos_token_id, pad_token_id=tok.pad_token_id)
    c2 = CLIPTextConfig(vocab_size=n_vocab, hidden_size=48, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2, projection_dim=40,
                        max_position_embeddings=77, eos_token_id=tok.eos_token_id, bos_token_id=tok.bos_token_id, pad_token_id=tok.pad_token_id)
    e1, e2 = CLIPTextModel(c1), CLIPTextModelWithProjection(c2)
    enc = ClipEncodersXL([tok, tok], [e1, e2], torch.device("cpu"))
    pe, ne, pp, npool = enc(["the cat", "the dog"], None)
    assert pe.shape == (2, 77, 80) and pp.shape == (2, 40) and not ne.any() and not npool.any()       # zeroed negative (xl.py:363-366)
    pe2, ne2, pp2, np2 = enc(["the cat", "the dog"], [""])
    assert torch.equal(pe, pe2) and ne2.shape == (1, 77, 80) and ne2.abs().sum() > 0 and np2.shape == (1, 40)
    ids = tok(["the cat"], padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
    with torch.no_grad():
        o1, o2 = e1(ids, output_hidden_states=True), e2(ids, output_hidden_states=True)
    assert torch.allclose(pe[:1], torch.cat([o1.hidden_states[-2], o2.hidden_states[-2]], -1)) and torch.allclose(pp[:1], o2[0])


def test_bpe_matches_the_hf_tokenizers_clip_recipe(tmp_path):
    """Independent pin for the BPE merge loop / byte alphabet / pre-tokenisation: the `tokenizers` library configured the way
    CLIPTokenizerFast is (NFC + whitespace collapse + lowercase, the CLIP split regex, byte-level alphabet, `</w>` suffix)."""
    from tokenizers import Regex, Tokenizer, normalizers, pre_tokenizers
    from tokenizers.models import BPE
    from rich_text_to_image_amd import clip_tokenizer as ct
    alpha = list(ct._byte_alphabet().values())
    vocab = alpha + [a + "</w>" for a in alpha]
    merges = [("c", "a"), ("ca", "t</w>"), ("t", "h"), ("th", "e</w>"), ("d", "o"), ("do", "g</w>"), ("i", "n"), ("in", "g</w>"), ("r", "u"),
              ("ru", "n"), ("n", "ing</w>"), ("a", "n"), ("an", "d</w>"), ("s", "k"), ("sk", "y</w>"), ("'", "s</w>")]
    vocab += [a + b for a, b in merges] + ["<|startoftext|>", "<|endoftext|>"]
    v = {t: i for i, t in enumerate(vocab)}
    ref = Tokenizer(BPE(vocab=v, merges=merges, end_of_word_suffix="</w>", unk_token="<|endoftext|>", continuing_subword_prefix="", fuse_unk=False))
    ref.normalizer = normalizers.Sequence([normalizers.NFC(), normalizers.Replace(Regex(r"\s+"), " "), normalizers.Lowercase()])
    pat = r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+"""
    ref.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.Split(Regex(pat), behavior="removed", invert=True),
                                                 pre_tokenizers.ByteLevel(add_prefix_space=False)])
    json.dump(v, open(tmp_path / "vocab.json", "w"))
    open(tmp_path / "merges.txt", "w").write("#version: 0.2\n" + "\n".join(a + " " + b for a, b in merges) + "\n")
    mine = ct.ClipBPETokenizer.from_pretrained(str(tmp_path))
    for txt in ["The cat's  running dog, 42 cats!", "a night sky and the dog's running", "café über naïve 東京", "it's the cats' sky...!!",
                "  multiple   spaces\tand\nnewlines ", "A close-up 4k dslr photo of a cat riding a scooter. There are palm trees in the background."]:
        assert mine._tokenize(txt) == ref.encode(txt).tokens, txt
        assert mine.convert_tokens_to_ids(mine._tokenize(txt)) == ref.encode(txt).ids


def test_checkpoint_config_readers(tmp_path):
    """unet/config.json and vae/config.json of a diffusers-layout checkpoint -> engine config dicts (absent keys keep diffusers' defaults)."""
    import os
    from rich_text_to_image_amd.checkpoint import unet_config, vae_config
    from rich_text_to_image_amd.engine import SD15_CONFIG, SD_VAE_CONFIG, SDXL_CONFIG
    os.makedirs(tmp_path / "unet"); os.makedirs(tmp_path / "vae")
    # SD-v1.5's config.json has no transformer_layers_per_block / use_linear_projection / addition_* keys
    json.dump({"_class_name": "UNet2DConditionModel", "in_channels": 4, "out_channels": 4, "block_out_channels": [320, 640, 1280, 1280],
               "down_block_types": ["CrossAttnDownBlock2D"] * 3 + ["DownBlock2D"], "up_block_types": ["UpBlock2D"] + ["CrossAttnUpBlock2D"] * 3,
               "layers_per_block": 2, "attention_head_dim": 8, "cross_attention_dim": 768, "norm_num_groups": 32, "sample_size": 64},
              open(tmp_path / "unet" / "config.json", "w"))
    cfg = unet_config(str(tmp_path / "unet"), SDXL_CONFIG)          # a wrong default must be fully overridden
    for k in ("block_out_channels", "down_block_types", "up_block_types", "cross_attention_dim", "layers_per_block", "norm_num_groups"):
        assert cfg[k] == SD15_CONFIG[k], k
    assert cfg["transformer_layers_per_block"] == 1 and cfg["use_linear_projection"] is False and cfg["addition_embed_type"] is None
    assert cfg["attention_head_dim"] == 8
    json.dump({"block_out_channels": [128, 256, 512, 512], "layers_per_block": 2, "norm_num_groups": 32, "scaling_factor": 0.13025},
              open(tmp_path / "vae" / "config.json", "w"))
    v = vae_config(str(tmp_path / "vae"), SD_VAE_CONFIG)
    assert v["scaling_factor"] == 0.13025 and v["block_out_channels"] == (128, 256, 512, 512)
    assert unet_config(str(tmp_path / "nowhere"), SD15_CONFIG) == SD15_CONFIG      # no config.json: the family default
    json.dump({"resnet_time_scale_shift": "scale_shift"}, open(tmp_path / "unet" / "config.json", "w"))
    with pytest.raises(ValueError):
        unet_config(str(tmp_path / "unet"), SD15_CONFIG)
