"""init_model(args, configs) -> (model, configs): the reference's model-assembly entry point
(wenet/utils/init_model.py:99-281) for the accelerated path.

Reads the same config keys (cmvn, cmvn_conf, input_dim, output_dim, encoder, encoder_conf, ctc, ctc_conf,
model, model_conf, tokenizer_conf, dataset_conf), honours args.checkpoint, and sets configs['init_infos'],
['num_seen_frames'], ['step'] and the model attributes lsl_enc / lsl_dec / add_cat_embs / cat_labels exactly like
the reference (:242-268).  It builds encoder + CTC (+ RNN predictor and joint for `model: transducer`); the `decoder` section of a reference
YAML is accepted and skipped (a checkpoint's extra keys are ignored the way the reference's def_strict=False does, :243).
"""
import logging

import torch

from ..transformer.asr_model import ASRModel
from ..transformer.cmvn import GlobalCMVN
from ..transformer.ctc import CTC
from ..transformer.encoder import ConformerEncoder
from .checkpoint import load_checkpoint
from .cmvn import load_cmvn

WENET_ENCODER_CLASSES = {"conformer": ConformerEncoder}
WENET_CTC_CLASSES = {"ctc": CTC}


def init_model(args, configs):
    dataset_conf = configs.get("dataset_conf", {})
    if configs.get("cmvn", None) == "global_cmvn":
        mean, istd = load_cmvn(configs["cmvn_conf"]["cmvn_file"], configs["cmvn_conf"]["is_json_cmvn"])
        global_cmvn = GlobalCMVN(torch.from_numpy(mean).float(), torch.from_numpy(istd).float())
    else:
        global_cmvn = None

    input_dim = configs["input_dim"]
    vocab_size = configs["output_dim"]
    encoder_type = configs.get("encoder", "conformer")
    if encoder_type not in WENET_ENCODER_CLASSES:
        raise NotImplementedError(f"encoder {encoder_type!r}: only the Conformer stack carries the recurrent slot")
    if dataset_conf.get("pass_cat_emb", False):
        raise NotImplementedError("language-specific layers are outside the accelerated path")
    configs["encoder_conf"]["num_langs"] = 0

    encoder = WENET_ENCODER_CLASSES[encoder_type](input_dim, global_cmvn=global_cmvn, **configs["encoder_conf"])
    ctc = WENET_CTC_CLASSES[configs.get("ctc", "ctc")](
        vocab_size, encoder.output_size(),
        blank_id=configs["ctc_conf"]["ctc_blank_id"] if "ctc_conf" in configs else 0)
    ctc.fp32_split_operands = bool(getattr(encoder, "fp32_split_operands", True))   # a pure-fp32 model keeps exact fp32 products
    model_conf = dict(configs.get("model_conf", {}))
    special = configs.get("tokenizer_conf", {}).get("special_tokens", None)
    if configs.get("model", "asr_model") == "transducer" and "predictor_conf" in configs and "joint_conf" in configs:
        # init_model.py:192-209: RNN predictor + joint around the same encoder / CTC (config c5)
        from ..transducer.joint import TransducerJoint
        from ..transducer.predictor import RNNPredictor
        from ..transducer.transducer import Transducer
        if configs.get("predictor", "rnn") != "rnn":
            raise NotImplementedError("only the RNN predictor of the paper's configs is implemented")
        predictor = RNNPredictor(vocab_size, **configs["predictor_conf"])
        joint = TransducerJoint(vocab_size, **configs["joint_conf"])
        model = Transducer(vocab_size=vocab_size, blank=0, encoder=encoder, predictor=predictor, joint=joint, ctc=ctc,
                           special_tokens=special, **model_conf)
    else:
        model = ASRModel(vocab_size=vocab_size, encoder=encoder, ctc=ctc, special_tokens=special, **model_conf)

    if getattr(args, "checkpoint", None) is not None:
        infos = load_checkpoint(model, args.checkpoint, def_strict=False)
    else:
        infos = {}
    configs["init_infos"] = infos
    configs["num_seen_frames"] = infos.get("num_seen_frames", 0)
    configs["step"] = infos.get("step", 0)

    model.lsl_enc = False
    model.lsl_dec = False
    model.add_cat_embs = dataset_conf.get("add_cat_emb", False)
    model.cat_labels = []
    logging.info("init_model: %d encoder parameters", sum(p.numel() for p in encoder.parameters()))
    return model, configs
