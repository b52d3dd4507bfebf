"""Debug aid: run bench.py under faulthandler so that a hang prints the host stack after N seconds."""
import faulthandler, runpy, sys
faulthandler.dump_traceback_later(int(sys.argv[1]), exit=True)
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path("bench.py", run_name="__main__")
