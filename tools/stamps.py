import sys, numpy as np
a=np.loadtxt(sys.argv[1])*0.01
names=['start','loaded','ranked','lookback_done','reordered','end']
print('tiles',len(a))
for k in range(6): print('%-14s min %6.2f med %6.2f max %6.2f'%(names[k], a[:,k].min(), np.median(a[:,k]), a[:,k].max()))
d=np.diff(a,axis=1)
for k in range(5): print('phase %-14s med %6.2f max %6.2f'%(names[k+1], np.median(d[:,k]), d[:,k].max()))
