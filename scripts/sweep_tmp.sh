cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import sys,os; sys.path.insert(0,'.')
import numpy as np
from raytracer_project_amd import capi
for u in (2,4,6,8):
    os.environ['ZR_STREAM_UNITS_PER_SLOT']=str(u)
    ctx=capi.Context(0); ds=capi.DemoScene('cfg3'); sc=capi.Scene(ctx,ds.desc)
    out=np.zeros((1080,1920,3))
    for mod in (8,4,1):
        reg=capi.Region(0,0,0,0,32,mod,0,0)
        sc.render(ds.camera,ds.env,ds.seed,reg,out=out,count=True); seg=ctx.counters().segments
        sc.render(ds.camera,ds.env,ds.seed,reg,out=out); c=ctx.counters()
        print('u',u,'1/%d of tiles'%mod, 'Mseg/s %.1f'%(seg/c.kernel_ms*1e-3), 'rounds',c.rounds, 'ms %.1f'%c.kernel_ms)
PY
for u in 4 8; do ZR_STREAM_UNITS_PER_SLOT=$u python scripts/stats.py cfg2:256 cfg5:1024 | grep Mseg | cut -c1-100; done
