cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python3 - <<'PY'
import sys; sys.path.insert(0,'.')
from raytracer_project_amd import capi
ctx=capi.Context(0)
for name,spp in (('cfg3',512),('cfg2',256),('cfg5',256)):
    ds=capi.DemoScene(name); cam=ds.camera.copy(); cam.samples_per_pixel=spp
    sc=capi.Scene(ctx,ds.desc); sc.render(cam,ds.env,ds.seed,None,count=True); c=ctx.counters()
    d=c.as_dict(); seg=d['segments']
    sc.render(cam,ds.env,ds.seed,None,count=False); c2=ctx.counters()
    print(name, 'Mseg/s %.1f'%(seg/c2.kernel_ms*1e-3), 'boxes/seg %.1f'%(d['nodes_tested']/seg), 'tri/seg %.2f'%(d['triangles_tested']/seg), 'sph/seg %.2f'%(d['spheres_tested']/seg),'cube/seg %.2f'%(d['cubes_tested']/seg), 'hit frac %.2f'%(d['hits']/seg))
    for ph in ('node','leaf','shade'):
        e=d[ph+'_execs']; l=d[ph+'_lanes']; print('   %-5s execs/seg*64 %.2f  avg lanes %.1f'%(ph, e*64/seg, l/max(e,1)))
PY
