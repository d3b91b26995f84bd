"""tools/mlp_only.py with the float16 decoder (argument-free target for tools/pmc_any.sh)."""
import os, runpy, sys
sys.argv = [sys.argv[0], "6", "f16"]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mlp_only.py"), run_name="__main__")
