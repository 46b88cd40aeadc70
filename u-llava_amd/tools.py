"""Host-side helpers of the model surface (integer / string work, no kernels).

Own restatement of reference `models/tools.py`: `KeywordsStoppingCriteria` (:11-31; the `generate()` stopping criterion the
inference scripts pass, `inference_ullava.py:92-101`) and the tokenizer / embedding growth helpers the training scripts call before
`Trainer.train()` (:34-117).  They only need `model.resize_token_embeddings`, `get_input_embeddings`, `get_output_embeddings`.
"""
from typing import Dict, List

import torch


class KeywordsStoppingCriteria:
    """Stop when batch row 0 ends with a single-token keyword or its decoded continuation contains a keyword string.
    Like the reference, the first call only records the prompt length (it is made after the first generated token)."""

    def __init__(self, keywords, tokenizer, input_ids):
        self.keywords = keywords
        ids = [tokenizer(k).input_ids for k in keywords]
        self.keyword_ids = [i[0] for i in ids if type(i) is list and len(i) == 1]
        self.tokenizer = tokenizer
        self.start_len = None
        self.input_ids = input_ids

    def __call__(self, output_ids: torch.LongTensor, scores: torch.FloatTensor = None, **kwargs) -> bool:
        if self.start_len is None:
            self.start_len = self.input_ids.shape[1]
            return False
        last = int(output_ids[0, -1])
        if any(last == k for k in self.keyword_ids):
            return True
        text = self.tokenizer.batch_decode(output_ids[:, self.start_len:], skip_special_tokens=True)[0]
        return any(k in text for k in self.keywords)


def _average_new_rows(model, num_new_tokens: int) -> None:
    if num_new_tokens <= 0:
        return
    for emb in (model.get_input_embeddings().weight.data, model.get_output_embeddings().weight.data):
        emb[-num_new_tokens:] = emb[:-num_new_tokens].mean(dim=0, keepdim=True)


def smart_special_token_and_embedding_resize(special_tokens_dict: Dict, tokenizer, model) -> None:
    """models/tools.py:34-58: add special tokens, grow the embeddings, new rows = mean of the old rows."""
    n = tokenizer.add_special_tokens(special_tokens_dict)
    model.resize_token_embeddings(len(tokenizer))
    _average_new_rows(model, n)


def smart_resize_token_embedding(new_tokens: List, tokenizer, model) -> None:
    """models/tools.py:61-86."""
    n = tokenizer.add_tokens(new_tokens, special_tokens=True)
    model.resize_token_embeddings(len(tokenizer))
    _average_new_rows(model, n)


def multi_modal_resize_token_embedding(mm_tokens: Dict, tokenizer, model) -> None:
    """models/tools.py:89-117: patch tokens are added without averaging, start / end tokens with it."""
    tokenizer.add_tokens([mm_tokens["IMG_PATCH"], mm_tokens["VID_PATCH"]], special_tokens=True)
    model.resize_token_embeddings(len(tokenizer))
    n = tokenizer.add_tokens([mm_tokens["IMG_START"], mm_tokens["IMG_END"], mm_tokens["VID_START"], mm_tokens["VID_END"]], special_tokens=True)
    model.resize_token_embeddings(len(tokenizer))
    _average_new_rows(model, n)
