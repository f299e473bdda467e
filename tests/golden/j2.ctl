GENERAL-INFO-START

	seq-file            j2.seq
	trace-file          j2.trace
	locus-mut-rate          CONST
	num-loci            10
	random-seed         12345
	mcmc-iterations	  100
	iterations-per-log  25
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
		samples		s2 d
	POP-END

	POP-START
		name		D
		samples		s3 d
	POP-END

	POP-START
		name		E
		samples		s4 d
	POP-END

	POP-START
		name		F
		samples		s5 d
	POP-END

CURRENT-POPS-END

ANCESTRAL-POPS-START

	POP-START
		name			BC
		children		B		C
		tau-initial	0.000004000
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			ABC
		children		A		BC
		tau-initial	0.000008560
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			DE
		children		D		E
		tau-initial	0.000004560
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			DEF
		children		DE		F
		tau-initial	0.000009680
		tau-beta		20000.0	
		finetune-tau			0.00000080
	POP-END

	POP-START
		name			root
		children		ABC		DEF
		tau-initial	0.000020480
		tau-beta		20000.0	
		finetune-tau			0.00000286
	POP-END

ANCESTRAL-POPS-END

MIG-BANDS-START	
	BAND-START		
       source  A
       target  BC
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  BC
       target  A
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  DE
       target  F
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  F
       target  DE
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  ABC
       target  DEF
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  DEF
       target  ABC
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  B
       target  C
       mig-rate-print 0.1
	BAND-END

	BAND-START		
       source  D
       target  BC
       mig-rate-print 0.1
	BAND-END

MIG-BANDS-END
