GENERAL-INFO-START

	seq-file            j3.seq
	trace-file          j3.trace
	locus-mut-rate          CONST
	num-loci            12
	random-seed         12345
	mcmc-iterations	  120
	iterations-per-log  40
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
	mig-rate-beta		0.0000000400

GENERAL-INFO-END

CURRENT-POPS-START	

	POP-START
		name		A
		samples		s0 d
	POP-END

	POP-START
		name		B
		samples		s1 d
	POP-END

	POP-START
		name		C
		samples		s2 d s3 d
		age		0.000002 e
	POP-END

	POP-START
		name		D
		samples		s4 d
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
		name			CD
		children		C		D
		tau-initial	0.000005350
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			root
		children		AB		CD
		tau-initial	0.000011400
		tau-beta		20000.0	
		finetune-tau			0.00000286
	POP-END

ANCESTRAL-POPS-END

MIG-BANDS-START	
	BAND-START		
       source  AB
       target  CD
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  CD
       target  AB
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  D
       target  C
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  A
       target  B
       mig-rate-print 0.1
	BAND-END

MIG-BANDS-END
