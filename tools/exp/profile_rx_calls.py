import cProfile, pstats, sys, time, numpy as np
sys.path.insert(0,'/root/repo')
import opticommpy_amd as oa
def bag(**kw):
    p=oa.parameters()
    for k,v in kw.items(): setattr(p,k,v)
    return p
N=1<<20; Fs=512e9
rng=np.random.default_rng(1)
Es=oa.to_device((rng.normal(size=(N,2))+1j*rng.normal(size=(N,2)))*0.02); Elo=oa.to_device(np.full(N,np.sqrt(8e-3),dtype=complex))
fe=dict(polRotation=np.pi/3,pdl=0,polDelay=3/32e9); pd=dict(B=32e9,ideal=True)
h=oa.lowPassFIR(25e9,Fs,255)
def work():
    for _ in range(50):
        s=oa.pdmCoherentReceiver(Es,Elo,bag(Fs=Fs,**fe),bag(Fs=Fs,**pd))
        s=oa.firFilter(h,s)
        s=oa.decimate(s,bag(SpSin=16,SpSout=2))
work()
t=time.perf_counter(); work(); print("per chain ms",(time.perf_counter()-t)/50*1e3)
cProfile.run('work()','/tmp/rx.prof')
pstats.Stats('/tmp/rx.prof').sort_stats('cumulative').print_stats(25)
