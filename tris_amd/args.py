"""Command-line surface of the Stage-1 scripts: every flag and default of the reference parser (args.py:3-98) is
accepted, declared here as a table.  `get_parser()` returns an argparse parser like the reference's."""
import argparse

# (flags, kwargs) -- grouped as in the reference: dataset, REFER, optimiser, training, evaluation, output, distributed,
# loss coefficients, model, CAM export, demo
_S, _I, _F = str, int, float
_FLAGS = [
    (("--dataset",), dict(default="refcoco")),
    (("--max_query_len",), dict(default=20, type=_I)),
    (("--negative_samples",), dict(default=0, type=_I)),
    (("--positive_samples",), dict(default=1, type=_I)),
    (("--bert_tokenizer",), dict(default="clip")),
    (("--refer_data_root",), dict(default="../../data/")),
    (("--splitBy",), dict(default="unc")),
    (("--spilt",), dict(default="val")),
    (("--pretrained_checkpoint",), dict(default=None, type=_S)),
    (("--lr",), dict(default=5e-5, type=_F)),
    (("--weight-decay", "--weight_decay"), dict(default=0.01, type=_F)),
    (("--lr_multi",), dict(default=0.1, type=_F)),
    (("--end_lr",), dict(default=1e-5, type=_F)),
    (("--power",), dict(default=1.0, type=_F)),
    (("--max_decay_steps",), dict(default=40, type=_I)),
    (("--batch_size",), dict(default=1, type=_I)),
    (("--epoch",), dict(default=30, type=_I)),
    (("--print-freq",), dict(default=100, type=_I)),
    (("--size",), dict(default=384, type=_I)),
    (("--resume",), dict(action="store_true")),
    (("--start_epoch",), dict(default=0, type=_I)),
    (("--gpu",), dict(default="0", type=_S)),
    (("--pseudo_path",), dict(default=None, type=_S)),
    (("--pretrain",), dict(default=None, type=_S)),
    (("--eval",), dict(action="store_true")),
    (("--test_split",), dict(default="val", type=_S)),
    (("--prms",), dict(action="store_true")),
    (("--eval_mode",), dict(default="cat", type=_S)),
    (("--visualize",), dict(action="store_true")),
    (("--dcrf",), dict(action="store_true")),
    (("--model_ema",), dict(action="store_true")),
    (("--consistency_type",), dict(default="mse", type=_S)),
    (("--scales",), dict(default=None, type=_S)),
    (("--output",), dict(default=None, type=_S)),
    (("--board_folder",), dict(default=None, type=_S)),
    (("--vis_out",), dict(default=None, type=_S)),
    (("--eval_vis_out",), dict(default=None, type=_S)),
    (("--pooling",), dict(default="gmp_gap", type=_S)),
    (("--distributed",), dict(action="store_true")),
    (("--world-size",), dict(default=1, type=_I)),
    (("--dist-url",), dict(default="env://")),
    (("--local_rank",), dict(default=0)),
    (("--attn_multi",), dict(default=0.1, type=_F)),
    (("--attn_multi_vis",), dict(default=0.1, type=_F)),
    (("--attn_multi_text",), dict(default=0.1, type=_F)),
    (("--w1",), dict(default=1, type=_F)), (("--w2",), dict(default=0, type=_F)), (("--w3",), dict(default=0, type=_F)),
    (("--w4",), dict(default=5, type=_F)), (("--w5",), dict(default=2, type=_F)),
    (("--FOCAL_P",), dict(default=3, type=_F)),
    (("--FOCAL_LAMBDA",), dict(default=0.01, type=_F)),
    (("--wr",), dict(default=5e-4, type=_F)),
    (("--backbone",), dict(default="clip-RN50", type=_S)),
    (("--hidden_dim",), dict(default=1024, type=_I)),
    (("--cam_save_dir",), dict(default=None, type=_S)),
    (("--name_save_dir",), dict(default=None, type=_S)),
    (("--save_cam",), dict(action="store_true")),
    (("--mode",), dict(default="clip", type=_S)),
    (("--img",), dict(default=None, type=_S)),
    (("--text",), dict(default=None, type=_S)),
]


def get_parser():
    parser = argparse.ArgumentParser(description="TRIS Stage-1 on MI355X (flag-compatible with the reference scripts)")
    for flags, kw in _FLAGS:
        parser.add_argument(*flags, **kw)
    return parser
