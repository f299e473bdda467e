GENERAL-INFO-START

	seq-file            e2e_100000.seq
	trace-file          e2e_100000.trace
	locus-mut-rate          CONST
	num-loci            100000
	random-seed         12345
	mcmc-iterations	  24
	iterations-per-log  100
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
		samples		s0 d s1 d
	POP-END

	POP-START
		name		B
		samples		s2 d s3 d
	POP-END

	POP-START
		name		C
		samples		s4 d s5 d
	POP-END

	POP-START
		name		D
		samples		s6 d
	POP-END

	POP-START
		name		E
		samples		s7 d
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
		name			ABC
		children		AB		C
		tau-initial	0.000010000
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABCD
		children		ABC		D
		tau-initial	0.000020000
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			root
		children		ABCD		E
		tau-initial	0.000100000
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

	BAND-START		
       source  B
       target  A
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  D
       target  C
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  E
       target  D
       mig-rate-print 0.1
	BAND-END

MIG-BANDS-END
