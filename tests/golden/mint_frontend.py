#!/usr/bin/env python3
"""Mint the text-side frontend fixtures from the REFERENCE's own functions (build container only).

    python tests/golden/mint_frontend.py

Imports /root/reference/CosyVoice/cosyvoice/utils/frontend_utils.py (needs only `re` and `regex`) and records, for a list of
input strings, what contains_chinese / replace_blank / replace_corner_mark / remove_bracket / spell_out_number /
is_only_punctuation / split_paragraph return -> tests/golden/frontend_text.json.  cli/frontend.py itself cannot be imported
here (onnxruntime, whisper, inflect, torchaudio are absent), so the dict assembly of frontend_zero_shot / instruct2 is pinned by
restated-logic tests only (tests/test_frontend_cpu.py says so).

Also writes fangyan_tts_amd/cli/cv3_special_tokens.txt: the additional special tokens CosyVoice3Tokenizer registers
(tokenizer/tokenizer.py:274-313), read out of the class's literal with `ast` - a vocabulary table the token ids depend on.
"""
import ast
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/CosyVoice"
sys.path.insert(0, REF)

TEXTS = [
    "你好世界", "今天天气不错，我们去公园散步吧。", "Hello world. This is a test! Is it? Yes; it is: fine",
    "面积是 25 m² 还是 3 m³ ？", "他说：“走吧。”然后就走了！真的吗？是的；没错、好的", "（注）【重要】`code`——结束",
    "a b 中 文 c d", "mixed 中文 and English text 混合", "I have 2 apples and 15 oranges in 2024", "。。。", "!?", "",
    "第一句。第二句！第三句？第四句；第五句：第六句、第七句.eighth?ninth!tenth;", "no final stop", "结尾没有句号",
    "He said \"stop.\" Then left.", "价格 12.5 元 - 很便宜，，、",
]
LONG_ZH = "这是一个比较长的句子用来测试分段逻辑是否正确。" * 9 + "短句。"
LONG_EN = "This sentence is used to check how the paragraph splitter packs sentences into segments. " * 8 + "Short one."


def main():
    from cosyvoice.utils import frontend_utils as fu
    fx = {"texts": TEXTS, "contains_chinese": [], "replace_corner_mark": [], "remove_bracket": [], "replace_blank": [],
          "is_only_punctuation": [], "spell_out_number": [], "split": []}

    class Words:                                   # a stand-in for inflect.engine(): only number_to_words is called
        @staticmethod
        def number_to_words(s):
            return "<" + s + ">"
    for t in TEXTS:
        fx["contains_chinese"].append(fu.contains_chinese(t))
        fx["replace_corner_mark"].append(fu.replace_corner_mark(t))
        fx["remove_bracket"].append(fu.remove_bracket(t))
        fx["is_only_punctuation"].append(fu.is_only_punctuation(t))
        fx["spell_out_number"].append(fu.spell_out_number(t, Words))
        try:
            fx["replace_blank"].append(fu.replace_blank(t))
        except IndexError:
            fx["replace_blank"].append(None)
    tok = lambda s: s.split()                      # a stand-in tokenizer: one token per blank-separated word
    for text, lang in [(t, "zh") for t in TEXTS if t] + [(t, "en") for t in TEXTS if t] + [(LONG_ZH, "zh"), (LONG_EN, "en")]:
        for (mx, mn, mg, comma) in ((80, 60, 20, False), (30, 10, 8, True), (12, 4, 3, False)):
            try:
                out = fu.split_paragraph(text, tok, lang, token_max_n=mx, token_min_n=mn, merge_len=mg, comma_split=comma)
            except IndexError:
                out = None
            fx["split"].append({"text": text, "lang": lang, "args": [mx, mn, mg, comma], "out": out})
    # the token table
    src = open(os.path.join(REF, "cosyvoice/tokenizer/tokenizer.py"), encoding="utf-8").read()
    tokens = None
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.ClassDef) and node.name == "CosyVoice3Tokenizer":
            for d in ast.walk(node):
                if isinstance(d, ast.Dict):
                    for k, v in zip(d.keys, d.values):
                        if isinstance(k, ast.Constant) and k.value == "additional_special_tokens":
                            tokens = [e.value for e in v.elts]
    assert tokens and len(tokens) == len(set(tokens)), "token table not found"
    with open(os.path.join(ROOT, "fangyan_tts_amd", "cli", "cv3_special_tokens.txt"), "w", encoding="utf-8") as f:
        f.write("\n".join(tokens) + "\n")
    fx["special_tokens"] = {"count": len(tokens), "sha256": hashlib.sha256("\n".join(tokens).encode("utf-8")).hexdigest(),
                            "first": tokens[:3], "last": tokens[-3:]}
    with open(os.path.join(HERE, "frontend_text.json"), "w", encoding="utf-8") as f:
        json.dump(fx, f, ensure_ascii=False, indent=1)
    print(f"{len(TEXTS)} texts, {len(fx['split'])} split cases, {len(tokens)} special tokens")


if __name__ == "__main__":
    main()
