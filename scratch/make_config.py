"""Build train-procgen-pytorch_amd/hyperparams/procgen/config.yml: the hyper-parameter sets of the reference's config.yml that the
accelerated path can run (algo: ppo with architecture impala | mlpmodel -- SURVEY.md section 2 rows 5b / 15 put every other
algo / architecture out of scope), keys and values unchanged.  Build-container only (reads /root/reference)."""
import os, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ref = yaml.safe_load(open("/root/reference/hyperparams/procgen/config.yml"))
keep = {k: v for k, v in ref.items() if isinstance(v, dict) and v.get("algo", "ppo") == "ppo" and v.get("architecture") in ("impala", "mlpmodel")}
dropped = sorted(set(ref) - set(keep))
head = ("# Hyper-parameter sets for train.py --param_name (PPO on the IMPALA-CNN / MLP embedders): the %d sets of the reference's\n"
        "# hyperparams/procgen/config.yml with `algo: ppo` and `architecture: impala | mlpmodel`, keys and values as there.\n"
        "# Not carried over (other agents / architectures, out of scope -- train.py raises NotImplementedError for them):\n#   %s\n"
        "# Keys PPO.__init__ does not know fall into **kwargs and are ignored, as in the reference (agents/ppo.py:39).\n") % (len(keep), ", ".join(dropped))
with open(os.path.join(ROOT, "train-procgen-pytorch_amd", "hyperparams", "procgen", "config.yml"), "w") as f:
    f.write(head)
    for k, v in keep.items():
        f.write("\n")
        yaml.safe_dump({k: v}, f, sort_keys=False, default_flow_style=False)
print(len(keep), "sets:", list(keep))
