"""`python -m text2human_amd.sample_from_pose -opt configs/sample_from_pose.yml`
(the reference's sample_from_pose.py entry point; see sample_from_parsing.py)."""
from .sample_from_parsing import run

if __name__ == '__main__':
    run(pose=True)
