"""Config 5 through the device pipeline with the prepare thread one, two or three items ahead, ordered or free (HELM_C5_LOOKAHEAD, HELM_C5_STRICT): the
job takes 2.14 s in every variant -- set-up and iterations compete for the same compute units."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for la, st in ((1, 1), (1, 0), (2, 0), (2, 1), (3, 0)):
    os.environ['HELM_C5_LOOKAHEAD'] = str(la); os.environ['HELM_C5_STRICT'] = str(st)
    out = bench.config5_leg(0)
    print('lookahead', la, 'strict', st, 'job %.3f s' % out['job_seconds'], '1e-10: %.3f' % out['job_seconds_rtol1e10'], flush=True)
    print('   ', [(w[:9], f, t) for w, f, t in out['pipelined_timeline_ms']], flush=True)
