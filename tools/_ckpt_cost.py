"""Dev: what does the per-epoch checkpoint of train_MuRCL cost (make_state = D2H copies, torch.save, best copy)?"""
import os, sys, time, tempfile, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from murcl_amd.train_MuRCL import build_parser, create_model, get_optimizer
from murcl_amd.models import rlmil
from murcl_amd.utils import checkpoint as C
dev = torch.device("cuda:0")
args = build_parser().parse_args(["--arch", "ABMIL", "--train_stage", "1"])
model, fc, ppo = create_model(args, 512, dev)
opt = get_optimizer(args, model, fc)
ppo = rlmil.PPO(512, args.model_dim, args.policy_hidden_dim, False, action_size=10)
d = tempfile.mkdtemp()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    st = C.make_state(1, model, fc, opt, ppo)
    t1 = time.time()
    C.save_checkpoint(st, False, d)
    t2 = time.time()
    C.save_checkpoint(st, True, d)
    t3 = time.time()
    print(f"make_state {1e3 * (t1 - t0):.1f} ms, save {1e3 * (t2 - t1):.1f} ms, save + best copy {1e3 * (t3 - t2):.1f} ms, file {os.path.getsize(os.path.join(d, 'checkpoint.pth.tar')) / 2**20:.1f} MiB", flush=True)
