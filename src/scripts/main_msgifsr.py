"""python -u scripts/main_msgifsr.py --dataset-dir ../datasets/<name>   (launched by start.sh)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from common import run  # noqa: E402

run('MSGIFSR')
