GENERAL-INFO-START

	seq-file            p6.seq
	trace-file          p6.trace
	locus-mut-rate          CONST
	num-loci            2
	random-seed         4242
	mcmc-iterations	  4
	iterations-per-log  2
	logs-per-line       10

	find-finetunes		FALSE
	finetune-coal-time	0.01		
	finetune-mig-time	0.3		
	finetune-theta		0.04
	finetune-mig-rate	0.02
	finetune-tau		0.0000008
	finetune-mixing		0.003

	tau-theta-print		10000.0
	tau-theta-alpha		1.0
	tau-theta-beta		10000.0

	mig-rate-print		0.001
	mig-rate-alpha		0.002
	mig-rate-beta		0.0000100000

GENERAL-INFO-END

CURRENT-POPS-START	

	POP-START
		name		A
		samples		s0 d s1 d s2 d s3 d s4 d s5 d
	POP-END

	POP-START
		name		B
		samples		s6 d s7 d s8 d s9 d s10 d s11 d
	POP-END

	POP-START
		name		C
		samples		s12 d s13 d s14 d s15 d s16 d s17 d
	POP-END

CURRENT-POPS-END

ANCESTRAL-POPS-START

	POP-START
		name			AB
		children		A		B
		tau-initial	0.000005000
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			root
		children		AB		C
		tau-initial	0.000025000
		tau-beta		20000.0	
		finetune-tau			0.00000286
	POP-END

ANCESTRAL-POPS-END

MIG-BANDS-START	
	BAND-START		
       source  A
       target  B
       mig-rate-print 0.1
	BAND-END

MIG-BANDS-END
