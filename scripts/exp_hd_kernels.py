"""Which fill kernel the two strands of the example loci get at flank 110 (three random flank sets each): HD 243-245 states, both
strands slot-major <4,4,2,1,false,0>; DM2 <4,5,4,1,false,4> (stacked) and <4,4,3,1,false,0>; (AAAT) lane-major <4,4,2,1,false,1>."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from warpstr_amd import synth
from warpstr_amd.caller import HipCaller
for pat in ['(AGC)AACAGCCGCCAC(CGC)', '((CAGG){CAGM})(CAGA)(CA)', '(AAAT)']:
    for seed in (1,2,3):
        locus = synth.make_locus(pat, 110, seed)
        hip = HipCaller([locus.template, locus.reverse], [110,110])
        print(pat, seed, locus.template.n_states, locus.reverse.n_states, hip.kernel_name(0), '|', hip.kernel_name(1))
        hip.close()
