import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "handsonvlm-release_amd"))
import torch
from hvlm.config import HandsOnVLMConfig
from hvlm.model import HandsOnVLMForCausalLM
from hvlm.synthetic import make_text_batch_np
cfg = HandsOnVLMConfig.vicuna_7b()
m = HandsOnVLMForCausalLM(cfg, device="cuda:0", training=False).init_random(0)
ids, _ = make_text_batch_np(1, 512, 0, n_hand=0); ids = torch.from_numpy(ids).cuda()
img = torch.randn(1, 100, 3, 224, 224, device="cuda").to(torch.bfloat16)
def T(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
tv, vis = T(lambda: m.visual_tokens(img))
tc, cache = T(lambda: m.stack.new_cache(1100))
tp, (lg, S) = T(lambda: m._prefill(ids, vis, None, cache))
td, _ = T(lambda: m._decode(17, None, S, cache), 20)
th, _ = T(lambda: m._decode(32100, torch.rand(1, 2, 2, device="cuda"), S, cache), 20)
ts, _ = T(lambda: torch.multinomial(torch.softmax(lg.float(), -1), 1).item(), 20)
print(json.dumps(dict(clip_ms=tv, cache_alloc_ms=tc, prefill_ms=tp, decode_ms=td, decode_hand_ms=th, sample_ms=ts, S=S)))
