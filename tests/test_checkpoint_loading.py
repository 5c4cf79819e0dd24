"""Local HF checkpoint directories (config.json + *.safetensors) through the product's loaders, on CPU: tiny random-init
`transformers` models are saved with `save_pretrained`, read back by `weights.load_*_safetensors_dir`, and the loaded
tensors must (a) carry the shapes / names the engines ask for and (b) reproduce the `transformers` forward when fed to the
numpy oracle -- i.e. the tensor-name mapping of every supported model family is right end to end."""
import json
import os

import numpy as np
import pytest

import verbatim_rag_amd  # noqa: F401
from oracle import bert_np as B
from oracle import modernbert_np as O
from verbatim_rag_amd.engine import BertShape, ModernBertShape, strip_prefix
from verbatim_rag_amd.weights import bert_canonical, load_bert_safetensors_dir, load_safetensors_dir

torch = pytest.importorskip("torch")
transformers = pytest.importorskip("transformers")

IDS = [1, 17, 45, 99, 3, 250, 7, 2]


def _bert_cfg(**kw):
    return dict(vocab_size=300, hidden_size=64, num_hidden_layers=2, num_attention_heads=1, intermediate_size=96,
                max_position_embeddings=40, type_vocab_size=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **kw)


def _jitter(model, seed):
    """LayerNorm gains / biases away from their 1 / 0 defaults so a swapped or dropped tensor shows."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.ndim == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g))


def test_bert_masked_lm_directory(tmp_path):
    torch.manual_seed(0)
    m = transformers.BertForMaskedLM(transformers.BertConfig(**_bert_cfg())).eval()
    _jitter(m, 1)
    m.save_pretrained(tmp_path, safe_serialization=True)
    shape, W, cfg = load_bert_safetensors_dir(str(tmp_path))
    assert isinstance(shape, BertShape) and (shape.model_type, shape.hidden_size, shape.num_hidden_layers, shape.vocab_size,
                                             shape.max_position_embeddings) == ("bert", 64, 2, 300, 40)
    assert W["l1.wqkv"].shape == (192, 64) and W["mlm.dec.b"].shape == (300,) and "mlm.dec.w" not in W   # decoder tied
    ocfg = B.BertConfig(vocab_size=300, hidden_size=64, num_hidden_layers=2, num_attention_heads=1, intermediate_size=96,
                        max_position_embeddings=40)
    with torch.no_grad():
        out = m(torch.tensor([IDS]), output_hidden_states=True)
    hid = B.encoder_forward(ocfg, W, IDS)
    assert np.abs(hid - out.hidden_states[-1][0].numpy()).max() < 2e-5
    assert np.abs(B.mlm_logits(ocfg, W, hid) - out.logits[0].numpy()).max() < 2e-4
    oracle_names = B.canonical_from_hf({k: v.numpy() for k, v in m.state_dict().items()})
    assert all(np.array_equal(W[k], oracle_names[k]) for k in W if k in oracle_names) and set(W) <= set(oracle_names) | {"emb.types"}


def test_distilbert_masked_lm_directory(tmp_path):
    torch.manual_seed(1)
    c = transformers.DistilBertConfig(vocab_size=300, dim=64, n_layers=2, n_heads=1, hidden_dim=96, max_position_embeddings=40,
                                      dropout=0.0, attention_dropout=0.0)
    m = transformers.DistilBertForMaskedLM(c).eval()
    _jitter(m, 2)
    m.save_pretrained(tmp_path, safe_serialization=True)
    shape, W, cfg = load_bert_safetensors_dir(str(tmp_path))
    assert (shape.model_type, shape.hidden_size, shape.intermediate_size, shape.num_hidden_layers) == ("distilbert", 64, 96, 2)
    assert "emb.type0" not in W
    ocfg = B.BertConfig(vocab_size=300, hidden_size=64, num_hidden_layers=2, num_attention_heads=1, intermediate_size=96,
                        max_position_embeddings=40)
    with torch.no_grad():
        out = m(torch.tensor([IDS]), output_hidden_states=True)
    hid = B.encoder_forward(ocfg, W, IDS)
    assert np.abs(hid - out.hidden_states[-1][0].numpy()).max() < 2e-5
    assert np.abs(B.mlm_logits(ocfg, W, hid) - out.logits[0].numpy()).max() < 2e-4


def test_bert_cross_encoder_directory(tmp_path):
    torch.manual_seed(2)
    m = transformers.BertForSequenceClassification(transformers.BertConfig(num_labels=1, **_bert_cfg())).eval()
    _jitter(m, 3)
    m.save_pretrained(tmp_path, safe_serialization=True)
    shape, W, cfg = load_bert_safetensors_dir(str(tmp_path))
    assert W["emb.types"].shape == (2, 64) and W["pooler.w"].shape == (64, 64) and W["cls.w"].shape == (1, 64)
    ocfg = B.BertConfig(vocab_size=300, hidden_size=64, num_hidden_layers=2, num_attention_heads=1, intermediate_size=96,
                        max_position_embeddings=40)
    types = [0, 0, 0, 0, 0, 1, 1, 1]
    with torch.no_grad():
        out = m(torch.tensor([IDS]), token_type_ids=torch.tensor([types]))
    hid = B.encoder_forward(ocfg, W, IDS, type_ids=types)
    assert np.abs(B.pair_logits(ocfg, W, hid) - out.logits[0].numpy()).max() < 2e-5


def test_bert_directory_errors(tmp_path):
    with open(tmp_path / "config.json", "w") as f:
        json.dump({"model_type": "roberta"}, f)
    with pytest.raises(ValueError, match="not bert / distilbert"):
        load_bert_safetensors_dir(str(tmp_path))
    with open(tmp_path / "config.json", "w") as f:
        json.dump({"model_type": "bert"}, f)
    with pytest.raises(FileNotFoundError):
        load_bert_safetensors_dir(str(tmp_path))
    with pytest.raises(KeyError):
        bert_canonical({"embeddings.word_embeddings.weight": np.zeros((3, 4), np.float32)})


def test_modernbert_qa_model_directory(tmp_path):
    """The reference's QAModel checkpoints name the encoder `bert.*` and add `classifier.*`
    (extractor_models/model.py:18,51,54)."""
    from safetensors.numpy import save_file

    torch.manual_seed(3)
    hc = transformers.ModernBertConfig(vocab_size=300, hidden_size=64, num_hidden_layers=4, num_attention_heads=1, intermediate_size=96,
                                       max_position_embeddings=256, pad_token_id=0, cls_token_id=1, sep_token_id=2, bos_token_id=1,
                                       eos_token_id=2, local_attention=8, global_attn_every_n_layers=3)
    m = transformers.ModernBertModel(hc).eval()
    _jitter(m, 4)
    sd = {"bert." + k: v.numpy() for k, v in m.state_dict().items()}
    rng = np.random.default_rng(0)
    sd["classifier.weight"], sd["classifier.bias"] = rng.standard_normal((2, 64)).astype(np.float32), rng.standard_normal(2).astype(np.float32)
    save_file(sd, str(tmp_path / "model.safetensors"))
    hc.save_pretrained(tmp_path)
    shape, tensors, cfg = load_safetensors_dir(str(tmp_path))
    assert isinstance(shape, ModernBertShape)
    assert (shape.hidden_size, shape.num_hidden_layers, shape.local_attention, shape.global_attn_every_n_layers, shape.sep_token_id) == (64, 4, 8, 3, 2)
    W = strip_prefix(tensors)
    assert "layers.0.attn_norm.weight" not in W and W["layers.3.mlp.Wi.weight"].shape == (192, 64) and W["classifier.weight"].shape == (2, 64)
    ocfg = O.EncoderConfig(vocab_size=300, hidden_size=64, num_hidden_layers=4, num_attention_heads=1, intermediate_size=96,
                           pad_token_id=0, cls_token_id=1, sep_token_id=2, local_attention=shape.local_attention,
                           global_attn_every_n_layers=3, global_rope_theta=shape.global_rope_theta, local_rope_theta=shape.local_rope_theta)
    ids = list(np.random.default_rng(1).integers(3, 300, 40))
    with torch.no_grad():
        ref = m(torch.tensor([ids])).last_hidden_state[0].numpy()
    assert np.abs(O.encoder_forward(ocfg, W, ids) - ref).max() < 5e-5
    assert os.path.exists(tmp_path / "config.json")


def test_modernbert_token_classification_directory(tmp_path):
    """HF task-model naming (`model.*`, `head.*`, `classifier.*`): what the v2 highlighter path loads
    (extractors.py:151-166 builds an AutoModel whose head is ModernBertForTokenClassification's)."""
    torch.manual_seed(4)
    hc = transformers.ModernBertConfig(vocab_size=300, hidden_size=64, num_hidden_layers=3, num_attention_heads=1, intermediate_size=96,
                                       max_position_embeddings=256, pad_token_id=0, cls_token_id=1, sep_token_id=2, bos_token_id=1,
                                       eos_token_id=2, local_attention=8, num_labels=2, classifier_dropout=0.0, mlp_dropout=0.0)
    m = transformers.ModernBertForTokenClassification(hc).eval()
    _jitter(m, 5)
    m.save_pretrained(tmp_path, safe_serialization=True)
    shape, tensors, cfg = load_safetensors_dir(str(tmp_path))
    for name in ("head.dense.weight", "head.norm.weight", "classifier.weight", "classifier.bias", "model.embeddings.tok_embeddings.weight"):
        assert name in tensors, name
    W = strip_prefix(tensors)
    ocfg = O.EncoderConfig(vocab_size=300, hidden_size=64, num_hidden_layers=3, num_attention_heads=1, intermediate_size=96,
                           pad_token_id=0, cls_token_id=1, sep_token_id=2, local_attention=shape.local_attention,
                           global_attn_every_n_layers=shape.global_attn_every_n_layers, global_rope_theta=shape.global_rope_theta,
                           local_rope_theta=shape.local_rope_theta)
    ids = list(np.random.default_rng(2).integers(3, 300, 30))
    with torch.no_grad():
        ref = m(torch.tensor([ids])).logits[0].numpy()
    hid = O.encoder_forward(ocfg, W, ids)
    got = O.token_logits(hid, W["head.dense.weight"], W["head.norm.weight"], W["classifier.weight"], W["classifier.bias"], ocfg.norm_eps)
    assert np.abs(got - ref).max() < 1e-4


def test_format_detection_on_saved_directories(tmp_path):
    from verbatim_rag_amd.extractors import GpuModelSpanExtractor as E

    cfg = {"model_type": "modernbert", "architectures": ["ModernBertModel"], "hidden_size": 64}
    json.dump(cfg, open(tmp_path / "config.json", "w"))
    assert E._detect_format(str(tmp_path)) == E._FORMAT_QA_MODEL
    cfg["auto_map"] = {"AutoModel": "modeling_highlighter.VerbatimHighlighterModel"}
    json.dump(cfg, open(tmp_path / "config.json", "w"))
    assert E._detect_format(str(tmp_path)) == E._FORMAT_HIGHLIGHTER


def make_qa_checkpoint_dir(path, seed=5):
    """A directory shaped like the reference's v1 QA checkpoints: `bert.*` ModernBERT tensors + `classifier.*`,
    config.json and tokenizer.json (TINY geometry of the GPU tests: head_dim 64).  Returns (hf model, classifier W, b)."""
    import shutil

    from safetensors.numpy import save_file

    torch.manual_seed(seed)
    hc = transformers.ModernBertConfig(vocab_size=512, hidden_size=128, num_hidden_layers=4, num_attention_heads=2, intermediate_size=192,
                                       max_position_embeddings=8192, pad_token_id=0, cls_token_id=1, sep_token_id=2, bos_token_id=1,
                                       eos_token_id=2)
    m = transformers.ModernBertModel(hc).eval()
    _jitter(m, seed)
    rng = np.random.default_rng(seed)
    Wc, bc = rng.standard_normal((2, 128)).astype(np.float32), rng.standard_normal(2).astype(np.float32)
    sd = {"bert." + k: v.numpy() for k, v in m.state_dict().items()}
    sd["classifier.weight"], sd["classifier.bias"] = Wc, bc
    save_file(sd, os.path.join(path, "model.safetensors"))
    hc.save_pretrained(path)
    shutil.copy(os.path.join(os.path.dirname(__file__), "golden", "tokenizer.json"), os.path.join(path, "tokenizer.json"))
    return m, Wc, bc


def test_extractor_constructor_from_a_model_directory(tmp_path, monkeypatch):
    """`GpuModelSpanExtractor(model_path)` -- the drop-in form of `ModelSpanExtractor(model_path=...)`: reads the
    directory, picks the format, hands the encoder tensors and the right head to the engine, loads the tokenizer.  The
    engine is replaced by a recorder here; tests/test_extractor_gpu.py builds the real one from the same directory."""
    from verbatim_rag_amd import engine as eng_mod
    from verbatim_rag_amd.extractors import GpuModelSpanExtractor

    made = []

    class Recorder:
        max_seqs, max_tokens, max_ranges, qa_labels, token_labels = 64, 8192, 1024, 0, 0

        def __init__(self, shape, weights, **kw):
            self.shape, self.weights, self.kw = shape, weights, kw
            made.append(self)

        def set_qa_head(self, w, b):
            self.qa_labels, self.head = len(b), (w, b)

        def set_token_head(self, *a):
            self.token_labels, self.head = 2, a

    monkeypatch.setattr(eng_mod, "EncoderEngine", Recorder)
    with pytest.raises(FileNotFoundError):
        GpuModelSpanExtractor(model_path=str(tmp_path / "missing"))
    m, Wc, bc = make_qa_checkpoint_dir(str(tmp_path))
    ext = GpuModelSpanExtractor(model_path=str(tmp_path), threshold=0.5, n_engines=2)
    assert ext._format == ext._FORMAT_QA_MODEL and len(made) == 2 and len(ext.engines) == 2
    rec = made[0]
    assert rec.shape.hidden_size == 128 and rec.shape.sep_token_id == 2 and rec.kw["max_seq_len"] == 512
    assert np.array_equal(rec.head[0], Wc) and np.array_equal(rec.head[1], bc)
    assert np.array_equal(strip_prefix(rec.weights)["layers.2.attn.Wqkv.weight"], m.state_dict()["layers.2.attn.Wqkv.weight"].numpy())
    assert ext._tok.sep_token_id == 2 and ext._tok.ids("tower", add_special_tokens=True, max_length=16)[0] == 1
    # the packer runs on the directory's tokenizer
    sents, samples = ext.pack_qa("Where is the tower?", ["The tower is tall. It is in paris."])
    assert sents == [["The tower is tall.", "It is in paris."]] and len(samples[0].sentence_boundaries) == 2


def test_extractor_constructor_from_a_highlighter_directory(tmp_path, monkeypatch):
    """v2 checkpoints: `auto_map` names a *Highlighter* class, tensors follow ModernBertForTokenClassification."""
    import shutil

    from verbatim_rag_amd import engine as eng_mod
    from verbatim_rag_amd.extractors import GpuModelSpanExtractor

    made = []

    class Recorder:
        max_seqs, max_tokens, max_ranges, qa_labels, token_labels = 64, 8192, 1024, 0, 0

        def __init__(self, shape, weights, **kw):
            self.shape, self.weights, self.kw = shape, weights, kw
            made.append(self)

        def set_qa_head(self, w, b):
            raise AssertionError("a highlighter checkpoint must not get the sentence head")

        def set_token_head(self, dense, norm, cls_w, cls_b):
            self.token_labels, self.head = len(cls_b), (dense, norm, cls_w, cls_b)

    monkeypatch.setattr(eng_mod, "EncoderEngine", Recorder)
    torch.manual_seed(6)
    hc = transformers.ModernBertConfig(vocab_size=512, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=192,
                                       max_position_embeddings=8192, pad_token_id=0, cls_token_id=1, sep_token_id=2, bos_token_id=1,
                                       eos_token_id=2, num_labels=2)
    m = transformers.ModernBertForTokenClassification(hc).eval()
    m.save_pretrained(tmp_path, safe_serialization=True)
    cfg = json.load(open(tmp_path / "config.json"))
    cfg["auto_map"] = {"AutoModel": "modeling_verbatim.VerbatimHighlighterModel"}
    json.dump(cfg, open(tmp_path / "config.json", "w"))
    shutil.copy(os.path.join(os.path.dirname(__file__), "golden", "tokenizer.json"), tmp_path / "tokenizer.json")
    ext = GpuModelSpanExtractor(model_path=str(tmp_path), threshold=0.5, max_length=1024)
    rec = made[0]
    assert ext._format == ext._FORMAT_HIGHLIGHTER and rec.kw["max_seq_len"] == 1024 and rec.token_labels == 2
    sd = m.state_dict()
    assert np.array_equal(rec.head[0], sd["head.dense.weight"].numpy()) and np.array_equal(rec.head[2], sd["classifier.weight"].numpy())
    windows, offsets, n_ctx = ext._encode_windows("Where is the tower?", "The tall iron tower is in paris. " * 400)
    assert len(windows) > 1 and all(len(w[0]) <= 1024 for w in windows) and n_ctx == len(offsets)


class _BertRecorder:
    """Stands in for BertEncoderEngine: keeps what the directory loader hands over."""
    max_seqs, max_tokens, max_ranges = 64, 8192, 1024

    def __init__(self, shape, weights, **kw):
        self.shape, self.weights, self.kw = shape, weights, kw
        self.max_seq_len = min(kw.get("max_seq_len", 512), shape.max_position_embeddings)
        self.has_mlm = "mlm.dense.w" in weights
        self.pair_labels = int(weights["cls.w"].shape[0]) if "cls.w" in weights and "pooler.w" in weights else 0


def _save_tokenizer(path):
    import shutil

    shutil.copy(os.path.join(os.path.dirname(__file__), "golden", "tokenizer.json"), os.path.join(path, "tokenizer.json"))


def test_provider_and_reranker_constructors_from_directories(tmp_path, monkeypatch):
    """`from_directory`: the local-files form of `SpladeProvider(model_name)` / `SentenceTransformersProvider(model_name)` /
    `SentenceTransformersReranker(model_name)` (embedding_providers.py:55-71,120-133, rerankers.py:109-134)."""
    from verbatim_rag_amd import engine as eng_mod
    from verbatim_rag_amd.embedding_providers import GpuDenseProvider, GpuSpladeProvider
    from verbatim_rag_amd.rerankers import GpuCrossEncoderReranker

    monkeypatch.setattr(eng_mod, "BertEncoderEngine", _BertRecorder)
    torch.manual_seed(7)
    splade_dir, dense_dir, ce_dir = tmp_path / "splade", tmp_path / "dense", tmp_path / "ce"
    transformers.BertForMaskedLM(transformers.BertConfig(**_bert_cfg())).save_pretrained(splade_dir, safe_serialization=True)
    transformers.DistilBertModel(transformers.DistilBertConfig(vocab_size=300, dim=64, n_layers=2, n_heads=1, hidden_dim=96,
                                                               max_position_embeddings=40)).save_pretrained(dense_dir, safe_serialization=True)
    transformers.BertForSequenceClassification(transformers.BertConfig(num_labels=1, **_bert_cfg())).save_pretrained(ce_dir, safe_serialization=True)
    for d in (splade_dir, dense_dir, ce_dir):
        _save_tokenizer(d)
    os.makedirs(dense_dir / "1_Pooling")
    json.dump({"word_embedding_dimension": 64, "pooling_mode_cls_token": False, "pooling_mode_mean_tokens": True}, open(dense_dir / "1_Pooling" / "config.json", "w"))

    sp = GpuSpladeProvider.from_directory(str(splade_dir), max_length=128)
    assert sp.get_dimension() == 300 and sp.engine.has_mlm and sp.max_length == 40 and sp.engine.shape.model_type == "bert"
    dn = GpuDenseProvider.from_directory(str(dense_dir))
    assert dn.pooling == "mean" and dn.get_dimension() == 64 and dn.engine.shape.model_type == "distilbert"
    assert GpuDenseProvider.from_directory(str(splade_dir)).pooling == "cls"               # no 1_Pooling directory: the default
    with pytest.raises(ValueError, match="no MLM head"):
        GpuSpladeProvider.from_directory(str(dense_dir))
    rr = GpuCrossEncoderReranker.from_directory(str(ce_dir), rerank_k=7)
    assert rr.rerank_k == 7 and rr.engine.pair_labels == 1
    with pytest.raises(ValueError, match="no pair head"):
        GpuCrossEncoderReranker.from_directory(str(dense_dir))
    json.dump({"model_type": "roberta"}, open(tmp_path / "config.json", "w"))
    with pytest.raises(ValueError, match="not bert / distilbert / modernbert"):
        GpuSpladeProvider.from_directory(str(tmp_path))
    # the provider tokenises with the directory's tokenizer, special ids resolved (tokenizer or config)
    ids = sp._encode(["the tall tower"])[0]
    assert ids[0] == sp._tok.cls_token_id and ids[-1] == sp._tok.sep_token_id and len(ids) >= 3


def test_modernbert_masked_lm_directory_gets_the_mlm_head(tmp_path, monkeypatch):
    from verbatim_rag_amd import engine as eng_mod
    from verbatim_rag_amd.embedding_providers import GpuSpladeProvider

    class Recorder:
        max_seqs, max_tokens, max_ranges, has_mlm = 64, 8192, 1024, False

        def __init__(self, shape, weights, **kw):
            self.shape, self.weights, self.max_seq_len = shape, weights, kw.get("max_seq_len", 512)

        def set_mlm_head(self, dense_w, norm_w, decoder_b, decoder_w=None, split_operands=True):
            self.has_mlm, self.head, self.split = True, (dense_w, norm_w, decoder_b, decoder_w), split_operands

    monkeypatch.setattr(eng_mod, "EncoderEngine", Recorder)
    torch.manual_seed(8)
    hc = transformers.ModernBertConfig(vocab_size=512, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=192,
                                       max_position_embeddings=8192, pad_token_id=0, cls_token_id=1, sep_token_id=2, bos_token_id=1, eos_token_id=2)
    m = transformers.ModernBertForMaskedLM(hc).eval()
    m.save_pretrained(tmp_path, safe_serialization=True)
    _save_tokenizer(tmp_path)
    sp = GpuSpladeProvider.from_directory(str(tmp_path))
    sd = m.state_dict()
    assert sp.engine.has_mlm and np.array_equal(sp.engine.head[0], sd["head.dense.weight"].numpy())
    assert sp.engine.split is True                                  # split operands unless the caller opts out
    assert np.array_equal(sp.engine.head[2], sd["decoder.bias"].numpy()) and sp.engine.head[3] is None      # decoder tied: not stored
    assert sp.get_dimension() == 512 and sp._tok.sep_token_id == 2
