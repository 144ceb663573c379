import sys; sys.path.insert(0,'/root/repo')
import numpy as np, rtvqa_amd
from rtvqa_amd import _native as N, synth
eng=rtvqa_amd.Engine(0)
fr=synth.s_natural(8,1080,1920,seed=1234)
rec=eng.complexity(fr,mask=N.M_EDGE)
print('natural steps/frame',rec['hyst_steps'], 'tiles/frame',17*30, 'weak',rec['edge_weak'],'strong',rec['edge_strong'])
fr=synth.s_noise(4,1080,1920,seed=1)
rec=eng.complexity(fr,mask=N.M_EDGE)
print('noise steps/frame',rec['hyst_steps'], 'weak',rec['edge_weak'],'strong',rec['edge_strong'])
